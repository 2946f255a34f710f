"""CPU ORACLE (test infrastructure, NOT product code) -- NumPy restatement of CALAMITY's hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product (``calamity_amd``) never routes through it.

PARITY UNPINNED: the reference (TensorFlow + pyuvdata + hera_filters) cannot be imported in the build
container (``ModuleNotFoundError: tensorflow`` -- an ordinary error, not a permission denial) and none of
its tests pins a number on this path.  What this file pins instead:

* the forward ops are restated one-for-one from the reference source (file:line cited per function),
  on the reference's own zero-padded ``(nvecs, ngrps, nbls, nfreqs)`` tensors;
* the hand-derived adjoints are checked against torch-CPU autograd of that forward and against central
  finite differences (tests/test_oracle.py);
* optimizer updates follow the documented Keras semantics (epsilon outside the bias correction);
* loop semantics (unrecorded first update, pre-update losses, tol, use_min) follow
  calibration.py:681-738 line by line.

All "reference" citations are into /root/reference/calamity/.
"""
import numpy as np

# --------------------------------------------------------------------------------------------------
# L0 math: calibration.py:1587-1656
# --------------------------------------------------------------------------------------------------


def fg_model(fg_r, fg_i, fg_comps):
    """calibration.py:1587-1590 -- broadcast-multiply then reduce over the vector axis."""
    vr = np.sum(fg_r * fg_comps, axis=0)
    vi = np.sum(fg_i * fg_comps, axis=0)
    return vr, vi


def data_model(g_r, g_i, fg_r, fg_i, fg_comps, ant0_inds, ant1_inds):
    """calibration.py:1593-1605 -- gather gains, G = g_i conj(g_j) in split form, m = G v."""
    gr0 = g_r[ant0_inds]
    gr1 = g_r[ant1_inds]
    gi0 = g_i[ant0_inds]
    gi1 = g_i[ant1_inds]
    grgr = gr0 * gr1
    gigi = gi0 * gi1
    grgi = gr0 * gi1
    gigr = gi0 * gr1
    vr, vi = fg_model(fg_r, fg_i, fg_comps)
    model_r = (grgr + gigi) * vr + (grgi - gigr) * vi
    model_i = (gigr - grgi) * vr + (grgr + gigi) * vi
    return model_r, model_i


def mse(model_r, model_i, data_r, data_i, wgts):
    """calibration.py:1608-1609."""
    return np.sum((np.square(data_r - model_r) + np.square(data_i - model_i)) * wgts)


def ant_inds_from_corr_inds(corr_inds):
    """calibration.py:577-594 -- nested (chunk, group, baseline) antenna index arrays."""
    ant0_inds, ant1_inds = [], []
    for chunk in corr_inds:
        ant0_inds.append(np.asarray([[cp[0] for cp in grp] for grp in chunk], dtype=np.int64))
        ant1_inds.append(np.asarray([[cp[1] for cp in grp] for grp in chunk], dtype=np.int64))
    return ant0_inds, ant1_inds


def mse_chunked(g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds):
    """calibration.py:1612-1620."""
    cal_loss = []
    for cnum in range(nchunks):
        model_r, model_i = data_model(g_r, g_i, fg_r[cnum], fg_i[cnum], fg_comps[cnum], ant0_inds[cnum], ant1_inds[cnum])
        cal_loss.append(mse(model_r, model_i, data_r[cnum], data_i[cnum], wgts[cnum]))
    return np.sum(np.stack(cal_loss))


def mse_chunked_sum_regularized(
    g_r, g_i, fg_r, fg_i, fg_comps, nchunks, data_r, data_i, wgts, ant0_inds, ant1_inds, prior_r_sum, prior_i_sum
):
    """calibration.py:1623-1656."""
    cal_loss, model_r_sum, model_i_sum = [], [], []
    for cnum in range(nchunks):
        model_r, model_i = data_model(g_r, g_i, fg_r[cnum], fg_i[cnum], fg_comps[cnum], ant0_inds[cnum], ant1_inds[cnum])
        model_r_sum.append(np.sum(model_r * wgts[cnum]))
        model_i_sum.append(np.sum(model_i * wgts[cnum]))
        cal_loss.append(mse(model_r, model_i, data_r[cnum], data_i[cnum], wgts[cnum]))
    return (
        np.sum(np.stack(cal_loss))
        + np.square(np.sum(np.stack(model_r_sum)) - prior_r_sum)
        + np.square(np.sum(np.stack(model_i_sum)) - prior_i_sum)
    )


def prior_sums(sky_model_r, sky_model_i, wgts):
    """calibration.py:619-625."""
    pr = np.sum(np.stack([np.sum(sky_model_r[c] * wgts[c]) for c in range(len(wgts))]))
    pi = np.sum(np.stack([np.sum(sky_model_i[c] * wgts[c]) for c in range(len(wgts))]))
    return pr, pi


# --------------------------------------------------------------------------------------------------
# Hand-derived adjoints of the above (what tf.GradientTape computes at calibration.py:664-666).
# "Gradient" of a complex z = x + iy means (dL/dx, dL/dy): re/im are independent real variables
# (calibration.py:596-603).
# --------------------------------------------------------------------------------------------------


def loss_and_grads(
    g_r, g_i, fg_r, fg_i, fg_comps, data_r, data_i, wgts, ant0_inds, ant1_inds, prior_r_sum=None, prior_i_sum=None
):
    """Loss of mse_chunked[_sum_regularized] and its gradient w.r.t. [g_r, g_i] + fg_r + fg_i.

    With r = d - m:  e = -2 w r  (+ 2 (S_r - P_r) w + i 2 (S_i - P_i) w  for the "sum" regulariser);
    gbar_v = conj(G) e;  grad c = A^T gbar_v (summed over the baselines and channels of a group);
    gbar_G = conj(v) e;  grad g_i += gbar_G g_j;  grad g_j += conj(gbar_G) g_i.
    """
    nchunks = len(fg_comps)
    reg = prior_r_sum is not None
    fwd = []
    loss = 0.0
    s_r = 0.0
    s_i = 0.0
    for c in range(nchunks):
        a0, a1 = ant0_inds[c], ant1_inds[c]
        gr0, gr1, gi0, gi1 = g_r[a0], g_r[a1], g_i[a0], g_i[a1]
        G_r = gr0 * gr1 + gi0 * gi1
        G_i = gi0 * gr1 - gr0 * gi1
        v_r, v_i = fg_model(fg_r[c], fg_i[c], fg_comps[c])
        m_r = G_r * v_r - G_i * v_i
        m_i = G_i * v_r + G_r * v_i
        loss = loss + np.sum((np.square(data_r[c] - m_r) + np.square(data_i[c] - m_i)) * wgts[c])
        if reg:
            s_r = s_r + np.sum(m_r * wgts[c])
            s_i = s_i + np.sum(m_i * wgts[c])
        fwd.append((G_r, G_i, v_r, v_i, m_r, m_i, gr0, gr1, gi0, gi1))
    if reg:
        loss = loss + np.square(s_r - prior_r_sum) + np.square(s_i - prior_i_sum)
    grad_g_r = np.zeros_like(g_r)
    grad_g_i = np.zeros_like(g_i)
    grad_fg_r, grad_fg_i = [], []
    for c in range(nchunks):
        G_r, G_i, v_r, v_i, m_r, m_i, gr0, gr1, gi0, gi1 = fwd[c]
        e_r = -2.0 * wgts[c] * (data_r[c] - m_r)
        e_i = -2.0 * wgts[c] * (data_i[c] - m_i)
        if reg:
            e_r = e_r + 2.0 * (s_r - prior_r_sum) * wgts[c]
            e_i = e_i + 2.0 * (s_i - prior_i_sum) * wgts[c]
        # gbar_v = conj(G) e
        gv_r = G_r * e_r + G_i * e_i
        gv_i = G_r * e_i - G_i * e_r
        grad_fg_r.append(np.sum(fg_comps[c] * gv_r[None], axis=(2, 3))[:, :, None, None])
        grad_fg_i.append(np.sum(fg_comps[c] * gv_i[None], axis=(2, 3))[:, :, None, None])
        # gbar_G = conj(v) e
        gG_r = v_r * e_r + v_i * e_i
        gG_i = v_r * e_i - v_i * e_r
        # ant0 (g_i): gbar_G * g_j ; ant1 (g_j): conj(gbar_G) * g_i
        np.add.at(grad_g_r, ant0_inds[c], gG_r * gr1 - gG_i * gi1)
        np.add.at(grad_g_i, ant0_inds[c], gG_r * gi1 + gG_i * gr1)
        np.add.at(grad_g_r, ant1_inds[c], gG_r * gr0 + gG_i * gi0)
        np.add.at(grad_g_i, ant1_inds[c], gG_r * gi0 - gG_i * gr0)
    return loss, grad_g_r, grad_g_i, grad_fg_r, grad_fg_i


# --------------------------------------------------------------------------------------------------
# Optimizers: Keras semantics (tf.optimizers.Adam / Adamax, reached through calibration.py:17-27,
# :571, :667).  Third-party formulae, not in /root/reference; epsilon sits OUTSIDE the bias correction.
# --------------------------------------------------------------------------------------------------

OPTIMIZER_DEFAULTS = dict(learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7)


class Adam:
    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.t = 0
        self.state = {}

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        t = self.t
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        for n, (g, var) in enumerate(grads_and_vars):
            if n not in self.state:
                self.state[n] = (np.zeros_like(var), np.zeros_like(var))
            m, v = self.state[n]
            m *= self.b1
            m += (1.0 - self.b1) * g
            v *= self.b2
            v += (1.0 - self.b2) * g * g
            var -= (lr_t * m / (np.sqrt(v) + self.eps)).astype(var.dtype)


class Adamax:
    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.t = 0
        self.state = {}

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        t = self.t
        lr_t = self.lr / (1.0 - self.b1 ** t)
        for n, (g, var) in enumerate(grads_and_vars):
            if n not in self.state:
                self.state[n] = (np.zeros_like(var), np.zeros_like(var))
            m, u = self.state[n]
            m *= self.b1
            m += (1.0 - self.b1) * g
            np.maximum(self.b2 * u, np.abs(g), out=u)
            var -= (lr_t * m / (u + self.eps)).astype(var.dtype)


class _SlotOptimizer:
    """Shared plumbing of the optimizers below: per-variable slot arrays created on first use, iteration count."""

    nslots = 2
    slot_init = (0.0, 0.0)

    def __init__(self):
        self.t = 0
        self.state = {}

    def slots(self, n, var):
        if n not in self.state:
            self.state[n] = tuple(np.full_like(var, v) for v in self.slot_init[: self.nslots])
        return self.state[n]


class SGD(_SlotOptimizer):
    """tf.keras.optimizers.SGD (OptimizerV2, ResourceApplyGradientDescent / ResourceApplyKerasMomentum):
    momentum == 0: var -= lr g; else accum = momentum accum - lr g, var += accum (nesterov: var += momentum accum - lr g)."""

    nslots = 1

    def __init__(self, learning_rate=0.01, momentum=0.0, nesterov=False):
        super().__init__()
        self.lr, self.momentum, self.nesterov = learning_rate, momentum, nesterov

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            if self.momentum == 0.0:
                var -= (self.lr * g).astype(var.dtype)
                continue
            (accum,) = self.slots(n, var)
            accum *= self.momentum
            accum -= self.lr * g
            var += ((self.momentum * accum - self.lr * g) if self.nesterov else accum).astype(var.dtype)


class RMSprop(_SlotOptimizer):
    """tf.keras.optimizers.RMSprop (OptimizerV2, centered=False): rms = rho rms + (1 - rho) g^2; without momentum
    var -= lr g / (sqrt(rms) + epsilon); with momentum the fused ResourceApplyRMSProp, whose epsilon sits INSIDE the
    root: mom = momentum mom + lr g / sqrt(rms + epsilon), var -= mom."""

    def __init__(self, learning_rate=0.001, rho=0.9, momentum=0.0, epsilon=1e-7):
        super().__init__()
        self.lr, self.rho, self.momentum, self.eps = learning_rate, rho, momentum, epsilon

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            mom, rms = self.slots(n, var)
            rms *= self.rho
            rms += (1.0 - self.rho) * g * g
            if self.momentum == 0.0:
                var -= (self.lr * g / (np.sqrt(rms) + self.eps)).astype(var.dtype)
            else:
                mom *= self.momentum
                mom += self.lr * g / np.sqrt(rms + self.eps)
                var -= mom.astype(var.dtype)


class Adagrad(_SlotOptimizer):
    """tf.keras.optimizers.Adagrad (OptimizerV2, ResourceApplyAdagradV2): accum += g^2 (from initial_accumulator_value),
    var -= lr g / (sqrt(accum) + epsilon)."""

    def __init__(self, learning_rate=0.001, initial_accumulator_value=0.1, epsilon=1e-7):
        super().__init__()
        self.lr, self.eps = learning_rate, epsilon
        self.slot_init = (0.0, initial_accumulator_value)

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            _, accum = self.slots(n, var)
            accum += g * g
            var -= (self.lr * g / (np.sqrt(accum) + self.eps)).astype(var.dtype)


class Adadelta(_SlotOptimizer):
    """tf.keras.optimizers.Adadelta (OptimizerV2, ResourceApplyAdadelta): accum = rho accum + (1 - rho) g^2;
    update = sqrt(accum_update + epsilon) / sqrt(accum + epsilon) g; var -= lr update;
    accum_update = rho accum_update + (1 - rho) update^2."""

    def __init__(self, learning_rate=0.001, rho=0.95, epsilon=1e-7):
        super().__init__()
        self.lr, self.rho, self.eps = learning_rate, rho, epsilon

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            accum_update, accum = self.slots(n, var)
            accum *= self.rho
            accum += (1.0 - self.rho) * g * g
            update = np.sqrt(accum_update + self.eps) / np.sqrt(accum + self.eps) * g
            var -= (self.lr * update).astype(var.dtype)
            accum_update *= self.rho
            accum_update += (1.0 - self.rho) * update * update


class Nadam(_SlotOptimizer):
    """tf.keras.optimizers.Nadam (OptimizerV2): momentum schedule mu_t = beta_1 (1 - 0.5 * 0.96^(0.004 t)), the running
    product of the mu's as bias correction of the first moment:
    g' = g / (1 - prod_t); m = beta_1 m + (1 - beta_1) g; m' = m / (1 - prod_t mu_{t+1}); v = beta_2 v + (1 - beta_2) g^2;
    v' = v / (1 - beta_2^t); var -= lr ((1 - mu_t) g' + mu_{t+1} m') / (sqrt(v') + epsilon)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        super().__init__()
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.m_schedule = 1.0

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        t = self.t
        mu_t = self.b1 * (1.0 - 0.5 * 0.96 ** (0.004 * t))
        mu_t1 = self.b1 * (1.0 - 0.5 * 0.96 ** (0.004 * (t + 1)))
        sched_new = self.m_schedule * mu_t
        sched_next = sched_new * mu_t1
        self.m_schedule = sched_new
        for n, (g, var) in enumerate(grads_and_vars):
            m, v = self.slots(n, var)
            g_prime = g / (1.0 - sched_new)
            m *= self.b1
            m += (1.0 - self.b1) * g
            m_prime = m / (1.0 - sched_next)
            v *= self.b2
            v += (1.0 - self.b2) * g * g
            v_prime = v / (1.0 - self.b2 ** t)
            var -= (self.lr * ((1.0 - mu_t) * g_prime + mu_t1 * m_prime) / (np.sqrt(v_prime) + self.eps)).astype(var.dtype)


# calibration.py:17-27 without Ftrl and the tensorflow-addons LAMB (KeyError, like any unknown name at :571)
class Ftrl(_SlotOptimizer):
    """tf.keras.optimizers.Ftrl (OptimizerV2, ResourceApplyFtrl / ResourceApplyFtrlV2 with l2 shrinkage): slots linear (0) and
    accumulator (initial_accumulator_value);  accum' = accum + g^2;  sigma = (accum'^-p - accum^-p) / lr  (p = learning_rate_power,
    -0.5: square roots);  linear += g + 2 l2_shrinkage var - sigma var;
    var = |linear| > l1 ? (sign(linear) l1 - linear) / (accum'^-p / lr + 2 (l2 + beta / (2 lr))) : 0."""

    def __init__(self, learning_rate=0.001, learning_rate_power=-0.5, initial_accumulator_value=0.1, l1_regularization_strength=0.0,
                 l2_regularization_strength=0.0, l2_shrinkage_regularization_strength=0.0, beta=0.0):
        super().__init__()
        self.lr, self.p = learning_rate, learning_rate_power
        self.l1, self.l2 = l1_regularization_strength, l2_regularization_strength + beta / (2.0 * learning_rate)
        self.l2s = l2_shrinkage_regularization_strength
        self.slot_init = (0.0, initial_accumulator_value)

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            linear, accum = self.slots(n, var)
            new_accum = accum + g * g
            pn, po = new_accum ** (-self.p), accum ** (-self.p)
            linear += g + 2.0 * self.l2s * var - (pn - po) / self.lr * var
            quad = pn / self.lr + 2.0 * self.l2
            var[...] = np.where(np.abs(linear) > self.l1, (np.sign(linear) * self.l1 - linear) / quad, 0.0).astype(var.dtype)
            accum[...] = new_accum


class LAMB(_SlotOptimizer):
    """tensorflow_addons.optimizers.LAMB (calibration.py:26): Adam moments with bias correction, then one trust ratio PER VARIABLE:
    m = beta_1 m + (1 - beta_1) g; v = beta_2 v + (1 - beta_2) g^2; update = (m / (1 - beta_1^t)) / (sqrt(v / (1 - beta_2^t)) + epsilon)
    + weight_decay_rate var; ratio = |var| > 0 and |update| > 0 ? |var| / |update| : 1 (2-norms over the variable);
    var -= ratio learning_rate update."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-6, weight_decay_rate=0.0):
        super().__init__()
        self.lr, self.b1, self.b2, self.eps, self.wd = learning_rate, beta_1, beta_2, epsilon, weight_decay_rate

    def apply_gradients(self, grads_and_vars):
        self.t += 1
        for n, (g, var) in enumerate(grads_and_vars):
            m, v = self.slots(n, var)
            m *= self.b1
            m += (1.0 - self.b1) * g
            v *= self.b2
            v += (1.0 - self.b2) * g * g
            update = (m / (1.0 - self.b1**self.t)) / (np.sqrt(v / (1.0 - self.b2**self.t)) + self.eps) + self.wd * var
            wn, un = np.sqrt(np.sum(np.square(var.astype(np.float64)))), np.sqrt(np.sum(np.square(update.astype(np.float64))))
            ratio = (wn / un if un > 0.0 else 1.0) if wn > 0.0 else 1.0
            var -= (ratio * self.lr * update).astype(var.dtype)


OPTIMIZERS = {"Adam": Adam, "Adamax": Adamax, "SGD": SGD, "RMSprop": RMSprop, "Adagrad": Adagrad, "Adadelta": Adadelta, "Nadam": Nadam,
              "Ftrl": Ftrl, "LAMB": LAMB}


# --------------------------------------------------------------------------------------------------
# L1 fit loop: calibration.py:447-738
# --------------------------------------------------------------------------------------------------


def fit_gains_and_foregrounds(
    g_r,
    g_i,
    fg_r,
    fg_i,
    data_r,
    data_i,
    wgts,
    fg_comps,
    corr_inds,
    use_min=False,
    tol=1e-14,
    maxsteps=10000,
    optimizer="Adamax",
    freeze_model=False,
    dtype=np.float64,
    n_profile_steps=0,
    sky_model_r=None,
    sky_model_i=None,
    model_regularization=None,
    **opt_kwargs,
):
    """Loop semantics of calibration.py:447-738 on NumPy arrays.

    * ``n_profile_steps`` profiled steps (:681-687) and the "graph build" step (:693) are real,
      unrecorded updates;
    * recorded loss k is evaluated before update k (:663-668, :700-701);
    * ``use_min`` snapshots the post-update parameters of the lowest-loss step (:702-710);
    * stop when ``step >= 1 and |l_k - l_{k-1}| < tol`` (:712-717);
    * ``freeze_model`` optimises the gains only and returns fg_r/fg_i untouched (:598-603, :730-732).
    """
    opt = OPTIMIZERS[optimizer](**opt_kwargs)  # unknown name -> KeyError (:571)
    fit_history = {"loss": []}
    min_loss = 9e99
    ant0_inds, ant1_inds = ant_inds_from_corr_inds(corr_inds)
    g_r = np.array(g_r, dtype=dtype)
    g_i = np.array(g_i, dtype=dtype)
    fg_comps = [np.asarray(a, dtype=dtype) for a in fg_comps]
    data_r = [np.asarray(a, dtype=dtype) for a in data_r]
    data_i = [np.asarray(a, dtype=dtype) for a in data_i]
    wgts = [np.asarray(a, dtype=dtype) for a in wgts]
    fg_r_in, fg_i_in = fg_r, fg_i
    fg_r = [np.array(a, dtype=dtype) for a in fg_r]
    fg_i = [np.array(a, dtype=dtype) for a in fg_i]
    nchunks = len(fg_comps)
    if model_regularization == "sum":
        prior_r_sum, prior_i_sum = prior_sums(
            [np.asarray(a, dtype=dtype) for a in sky_model_r], [np.asarray(a, dtype=dtype) for a in sky_model_i], wgts
        )
    else:
        prior_r_sum = prior_i_sum = None

    def train_step():
        loss, gg_r, gg_i, gf_r, gf_i = loss_and_grads(
            g_r, g_i, fg_r, fg_i, fg_comps, data_r, data_i, wgts, ant0_inds, ant1_inds, prior_r_sum, prior_i_sum
        )
        if freeze_model:
            opt.apply_gradients([(gg_r, g_r), (gg_i, g_i)])
        else:
            opt.apply_gradients([(gg_r, g_r), (gg_i, g_i)] + list(zip(gf_r, fg_r)) + list(zip(gf_i, fg_i)))
        return dtype(loss) if isinstance(dtype, type) else loss

    for _ in range(n_profile_steps):
        train_step()
    train_step()
    g_r_opt = g_i_opt = fg_r_opt = fg_i_opt = None
    for step in range(maxsteps):
        loss = train_step()
        fit_history["loss"].append(loss)
        if use_min and fit_history["loss"][-1] < min_loss:
            min_loss = fit_history["loss"][-1]
            g_r_opt, g_i_opt = g_r.copy(), g_i.copy()
            if not freeze_model:
                fg_r_opt = [a.copy() for a in fg_r]
                fg_i_opt = [a.copy() for a in fg_i]
        if step >= 1 and np.abs(fit_history["loss"][-1] - fit_history["loss"][-2]) < tol:
            break
    if not use_min:
        g_r_opt, g_i_opt = g_r.copy(), g_i.copy()
        if not freeze_model:
            fg_r_opt = [a.copy() for a in fg_r]
            fg_i_opt = [a.copy() for a in fg_i]
        else:
            fg_r_opt, fg_i_opt = fg_r_in, fg_i_in
    elif freeze_model:
        # reference: fg_*_opt unbound here (latent UnboundLocalError, SURVEY 8a-8 v); return inputs.
        fg_r_opt, fg_i_opt = fg_r_in, fg_i_in
    return g_r_opt, g_i_opt, fg_r_opt, fg_i_opt, fit_history


# --------------------------------------------------------------------------------------------------
# Callers either side of the loop
# --------------------------------------------------------------------------------------------------


def yield_fg_model_array(nants, nfreqs, fg_model_comps, fg_coeffs, corr_inds):
    """calibration.py:402-444 -- post-fit A c of ONE real component into a (nants, nants, nfreqs) cube."""
    model = np.zeros((nants, nants, nfreqs))
    for cnum in range(len(fg_model_comps)):
        ngrps = fg_model_comps[cnum].shape[1]
        gchunk = np.sum(fg_coeffs[cnum] * fg_model_comps[cnum], axis=0)
        for gnum in range(ngrps):
            for blnum, (i, j) in enumerate(corr_inds[cnum][gnum]):
                model[i, j] = gchunk[gnum, blnum]
    return model


def tensorize_fg_coeffs(data, wgts, fg_model_comps):
    """calibration.py:828-913 -- per group least squares of ONE real component on the data with flagged
    samples zeroed (binary weights), over the non-padded vectors only, zero-padded back to nvecs."""
    fg_coeffs = []
    for cnum in range(len(data)):
        binary_wgts = (~np.isclose(wgts[cnum], 0.0)).astype(wgts[cnum].dtype)
        nvecs, ngrps = fg_model_comps[cnum].shape[:2]
        ndata = data[cnum].shape[1] * data[cnum].shape[2]
        chunk = np.zeros((nvecs, ngrps), dtype=data[cnum].dtype)
        for gnum in range(ngrps):
            amat = np.asarray(fg_model_comps[cnum][:, gnum]).reshape(nvecs, ndata)
            zero_rows = np.where(np.all(np.isclose(amat, 0.0), axis=1))[0]
            nvecs_nonzero = np.min(zero_rows) if len(zero_rows) > 0 else nvecs
            rhs = (data[cnum][gnum] * binary_wgts[gnum]).reshape(ndata)
            # tf.linalg.lstsq(fast=True): Cholesky on the normal equations.
            at = amat[:nvecs_nonzero].astype(np.float64)
            sol = np.linalg.solve(at @ at.T, at @ rhs.astype(np.float64))
            chunk[:nvecs_nonzero, gnum] = sol
        fg_coeffs.append(chunk.reshape(nvecs, ngrps, 1, 1))
    return fg_coeffs
