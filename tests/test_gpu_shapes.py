"""Shape coverage of the HIP kernels against the C restatement: every tile width (fp32 FB 128/64/32/16/8, fp64 FB 64/32/16/8/4
follow from nvec), nvec on both sides of every boundary, a channel count that is not a multiple of the tile width,
fitting groups of 1, 3 and 20 baselines (the latter flush the gbar_G buffer inside an item), one-baseline-per-group
problems for the dense (MFMA) path, and items split over workgroups."""
import numpy as np
import pytest

from calamity_amd.problem import FitProblem
from oracle.ref_c import CRef

pytestmark = pytest.mark.gpu


def random_problem(nvecs, bls_per_grp, nants=9, nfreqs=200, seed=0, rowblocks=False):
    rng = np.random.default_rng(seed)
    basis, grp_basis, start, a0, a1, rb = [], [], [0], [], [], []
    for n, (nvec, nb) in enumerate(zip(nvecs, bls_per_grp)):
        nrb = min(nb, 3) if rowblocks else 1
        basis.append(rng.standard_normal((nrb * nfreqs, nvec)) / np.sqrt(nfreqs))
        grp_basis.append(n)
        for b in range(nb):
            i, j = rng.choice(nants, size=2, replace=False)
            a0.append(i)
            a1.append(j)
            rb.append(b % nrb)
        start.append(start[-1] + nb)
    nbls = len(a0)
    w = rng.uniform(0.0, 1.0, size=(nbls, nfreqs)) * (rng.random((nbls, nfreqs)) > 0.1)
    p = FitProblem(nants=nants, nfreqs=nfreqs, basis=basis, grp_basis=np.asarray(grp_basis, np.int32), grp_bl_start=np.asarray(start, np.int32),
                   bl_ant0=np.asarray(a0, np.int32), bl_ant1=np.asarray(a1, np.int32), bl_rowblk=np.asarray(rb, np.int32),
                   data_r=rng.standard_normal((nbls, nfreqs)), data_i=rng.standard_normal((nbls, nfreqs)), wgts=w / w.sum())
    p.sky_r, p.sky_i = rng.standard_normal((nbls, nfreqs)), rng.standard_normal((nbls, nfreqs))
    start = dict(g_r=1.0 + 0.1 * rng.standard_normal((nants, nfreqs)), g_i=0.1 * rng.standard_normal((nants, nfreqs)),
                 c_r=rng.standard_normal(p.ncoeffs), c_i=rng.standard_normal(p.ncoeffs))
    return p, start


def check(p, start, dtypes=(np.float64, np.float32), layouts=("stream", "shared"), regs=(False, True), kernel_path="auto"):
    from calamity_amd.solver import HipFitSolver

    c = CRef(p, np.float64)
    for reg in regs:
        pr, pi = (float(np.sum(p.sky_r * p.wgts)), float(np.sum(p.sky_i * p.wgts))) if reg else (0.0, 0.0)
        c.set_regularization("sum" if reg else None, pr, pi)
        ref = c.loss_grads(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
        for dtype in dtypes:
            tol_l, tol_g = (1e-10, 1e-10) if dtype == np.float64 else (1e-5, 1e-4)
            for layout in layouts:
                s = HipFitSolver(dtype=dtype)
                s.set_problem(p, layout=layout, kernel_path=kernel_path)
                s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
                if reg:
                    s.set_regularization("sum", pr, pi)
                assert abs(s.eval_loss() - ref[0]) <= tol_l * abs(ref[0]), (dtype, layout, reg)
                out = s.eval_grads()
                assert abs(out[0] - ref[0]) <= tol_l * abs(ref[0]), (dtype, layout, reg)
                for a, b in zip(out[1:], ref[1:]):
                    assert np.linalg.norm(np.asarray(a, np.float64) - b) <= tol_g * np.linalg.norm(b), (dtype, layout, reg)
                s.close()


def test_every_tile_width_and_group_size():
    nvecs = [1, 7, 56, 57, 112, 113, 224, 225, 300, 448, 449, 896, 30, 100]
    bls = [1, 3, 20, 1, 3, 20, 2, 1, 3, 1, 2, 1, 20, 20]
    p, start = random_problem(nvecs, bls, seed=1)
    check(p, start)


def test_row_blocks_inside_groups():
    p, start = random_problem([5, 60, 130], [4, 7, 20], seed=2, rowblocks=True)
    check(p, start)


def test_dense_path_vector_counts():
    """One baseline per group, nvec <= 256, shared layout: the dense (matrix-core) kernels -- v_mfma_f32_32x32x2 in fp32
    (vector tiles of 32, panels of 16 baselines), v_mfma_f64_16x16x4 in fp64 (tiles of 16, panels of 8 or 16) -- across
    every tile-count class and tail; requested explicitly, problems below ~2000 baselines normally take the general kernel."""
    nvecs = [1, 7, 8, 9, 31, 32, 33, 64, 100, 129, 224, 255, 256] * 3
    p, start = random_problem(nvecs, [1] * len(nvecs), nants=12, nfreqs=1024, seed=3)
    # make same-shape groups share one basis block, as the operator cache does: panels of several baselines
    first = {}
    for g, n in enumerate(nvecs):
        p.grp_basis[g] = first.setdefault(n, g)
    check(p, start, dtypes=(np.float32, np.float64), layouts=("shared",), kernel_path="dense")
    check(p, start, dtypes=(np.float32,), layouts=("shared",), kernel_path="dense_f32")
    check(p, start, dtypes=(np.float32,), layouts=("shared",), kernel_path="dense_split1")


@pytest.mark.parametrize("path", ["dense", "dense_split1"])
def test_split_bf16_dense_kernel_vector_counts(path):
    """fp32, blocks of at most 224 vectors: the split-bf16 kernels -- "dense": split2_kernels.hpp (super-panels of 64 baselines, one operand
    image per unit of 64 / 32 channels read row-wise and transposed, a body per number of 32-vector tiles, coefficients in registers);
    "dense_split1": split_kernels.hpp (forward groups of two 16-vector steps, adjoint groups of two 32-vector tiles) -- across every
    tile count, tail and parity, both image shapes (blocks up to and beyond 128 vectors), with panels that are full, ragged and empty (a
    block with 70 baselines = one full panel of a super-panel and one of 6), with and without the regulariser; and that the path asked for
    is the path that ran."""
    from calamity_amd.solver import HipFitSolver

    nvecs = [1, 7, 8, 9, 15, 16, 17, 31, 32, 33, 48, 64, 65, 96, 100, 128, 129, 160, 161, 192, 193, 223, 224]
    reps = [1, 3, 70, 2, 17, 1, 16, 5, 64, 65, 2, 1, 33, 4, 20, 1, 18, 2, 1, 3, 1, 2, 1]
    allv = [n for n, r in zip(nvecs, reps) for _ in range(r)]
    p, start = random_problem(allv, [1] * len(allv), nants=40, nfreqs=1024, seed=5)
    first = {}
    for g, n in enumerate(allv):
        p.grp_basis[g] = first.setdefault(n, g)
    s = HipFitSolver(dtype=np.float32)
    s.set_problem(p, layout="shared", kernel_path=path)
    assert s.timing_get()["kernel_path"] == path
    s.close()
    check(p, start, dtypes=(np.float32,), layouts=("shared",), kernel_path=path)
    # a channel count that is not a multiple of the kernel's 64-channel pairs (zero-weight padding up to 128 k)
    p, start = random_problem([5, 40, 100] * 6, [1] * 18, nants=8, nfreqs=200, seed=6)
    first = {}
    for g, n in enumerate([5, 40, 100] * 6):
        p.grp_basis[g] = first.setdefault(n, g)
    check(p, start, dtypes=(np.float32,), layouts=("shared",), kernel_path=path)


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_split_bf16_dense_kernel_random_problems(seed):
    """Seeded random problems for the fp32 dense kernel of the SHARED layout (split2_kernels.hpp): 12 basis blocks of random width (1 .. 224
    vectors: every class of the kernel, both image shapes in one launch), random numbers of baselines per block (super-panels that are full,
    ragged or a single baseline), a band that is or is not a multiple of the 64-channel unit -- loss and gradients against the C restatement,
    with and without the regulariser."""
    rng = np.random.default_rng(seed)
    widths = [int(v) for v in rng.integers(1, 225, size=12)]
    counts = [int(v) for v in rng.choice([1, 2, 15, 16, 17, 63, 64, 65, 90], size=12)]
    nfreqs = int(rng.choice([130, 200, 513, 1024]))
    allv = [n for n, r in zip(widths, counts) for _ in range(r)]
    p, start = random_problem(allv, [1] * len(allv), nants=30, nfreqs=nfreqs, seed=100 + seed)
    pos = 0
    for r in counts:  # the baselines of a drawn block share ONE basis block (equal widths drawn twice stay different blocks)
        p.grp_basis[pos : pos + r] = pos
        pos += r
    check(p, start, dtypes=(np.float32,), layouts=("shared",), kernel_path="dense")


def test_many_channels_few_groups_split_items():
    """Fewer groups than workgroup slots: items are split by tiles and their coefficient gradients summed afterwards."""
    p, start = random_problem([20, 90, 250], [2, 5, 3], nants=6, nfreqs=4096, seed=4)
    check(p, start, regs=(False,))


def test_more_vectors_than_a_tile_holds_is_reported():
    from calamity_amd._lib import CalamityHipError
    from calamity_amd.solver import HipFitSolver

    p, start = random_problem([897], [1], seed=5)
    s = HipFitSolver(dtype=np.float32)
    with pytest.raises(CalamityHipError) as e:
        s.set_problem(p, layout="stream")
    assert e.value.code == -5 and "897" in str(e.value)
    s.close()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_model_and_init_on_multi_baseline_groups(dtype):
    """cal_solver_model (yield_fg_model_array, calibration.py:402-444) and cal_solver_init_coeffs (the A^T (d [w != 0]) part
    of tensorize_fg_coeffs, :875-902) for groups of many baselines, with and without several row blocks."""
    from calamity_amd.solver import HipFitSolver

    tol = 1e-11 if dtype == np.float64 else 2e-5
    for rowblocks in (False, True):
        p, start = random_problem([5, 60, 130, 300], [4, 7, 70, 20], seed=6, rowblocks=rowblocks)
        F = p.nfreqs
        coff = p.grp_coff
        for layout in ("stream", "shared"):
            s = HipFitSolver(dtype=dtype)
            s.set_problem(p, layout=layout)
            s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
            m_r, m_i = s.model()
            s.init_coeffs(p.data_r, p.data_i)
            _, _, c_r, c_i = s.get_params()
            for g in range(p.ngrps):
                blk = p.basis[p.grp_basis[g]]
                cg = start["c_r"][coff[g]:coff[g + 1]] + 1j * start["c_i"][coff[g]:coff[g + 1]]
                acc = np.zeros(blk.shape[1], dtype=np.complex128)
                for b in range(p.grp_bl_start[g], p.grp_bl_start[g + 1]):
                    A = blk[p.bl_rowblk[b] * F:(p.bl_rowblk[b] + 1) * F]
                    v = A @ cg
                    assert np.linalg.norm((m_r[b] + 1j * m_i[b]) - v) <= tol * max(np.linalg.norm(v), 1e-30)
                    msk = ~np.isclose(p.wgts[b], 0.0)
                    acc += A.T @ ((p.data_r[b] + 1j * p.data_i[b]) * msk)
                got = c_r[coff[g]:coff[g + 1]] + 1j * c_i[coff[g]:coff[g + 1]]
                assert np.linalg.norm(got - acc) <= tol * np.linalg.norm(acc)
            s.close()


@pytest.mark.parametrize("nslices", [3, 9, 10])
def test_time_slices_that_share_tiles(nslices):
    """cal_problem_desc::bl_alias: the same baselines in several time slices fitted by one solver read ONE copy of their basis
    tiles, and fused_multi_mfma_kernel (at most 224 vectors) / fused_multi_kernel process a baseline's slices together
    (SURVEY.md section 8e "Multiple times"; calibration.py:1160-1167 loops over times).  Every tile width, sets larger than a multi
    item holds (10 slices: 8 + 2 on the matrix-core kernel in both precisions, 4 + 4 + 2 for the 230-vector block in float64; 9 slices: the last baseline of a set is left over and runs as an
    ordinary item), loss / gradients against the C restatement of the batched problem, the regularised form
    (two passes of the same kernels: S first, then the gradients with every slice's alpha), model evaluation and initial coefficients, and a short trajectory against the one
    of the same problem WITHOUT the alias table (every baseline streaming its own copy; the loss partials are summed in another
    order, so equal to rounding, not to the bit)."""
    from calamity_amd import distributed as D
    from calamity_amd.solver import HipFitSolver

    nvecs = [5, 56, 57, 112, 113, 224, 230]  # FB 128 | 64 | 32 | 16 in float32
    base, _ = random_problem(nvecs, [1] * len(nvecs), nants=8, nfreqs=200, seed=40)
    parts = []
    for t in range(nslices):
        p, st = random_problem(nvecs, [1] * len(nvecs), nants=8, nfreqs=200, seed=41 + t)
        p.basis, p.grp_basis = base.basis, base.grp_basis          # the same blocks ...
        p.bl_ant0, p.bl_ant1, p.bl_rowblk = base.bl_ant0, base.bl_ant1, base.bl_rowblk  # ... and baselines in every slice
        p.wgts = p.wgts / nslices
        parts.append((p, st))
    prob, start = D.batch_time_slices(parts)
    assert prob.bl_alias is not None and np.all(prob.bl_alias[: base.nbls] == -1) and np.all(prob.bl_alias[base.nbls :] >= 0)
    prob.sky_r, prob.sky_i = np.concatenate([p.sky_r for p, _ in parts]), np.concatenate([p.sky_i for p, _ in parts])
    check(prob, start, layouts=("stream",))
    plain = D.batch_time_slices(parts)[0]
    plain.bl_alias = None
    pri = float(np.sum(prob.sky_r * prob.wgts)), float(np.sum(prob.sky_i * prob.wgts))
    for dtype, reg in ((np.float64, False), (np.float32, False), (np.float64, True), (np.float32, True)):
        # (with the regulariser: the two-pass form of the multi-slice kernels against the one-pass, two-adjoint-set form of the
        # same problem without the alias table)
        outs = []
        for pr in (prob, plain):
            s = HipFitSolver(dtype=dtype)
            s.set_problem(pr, layout="stream")
            mem = s.memory_bytes()
            s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
            if reg:
                s.set_regularization("sum", *pri)
            s.set_optimizer("Adam", learning_rate=1e-2)
            losses, _, _ = s.run(7, record=True, tol=0.0)
            model = s.model()
            s.init_coeffs(pr.data_r, pr.data_i)
            outs.append((losses, s.get_params(), model, mem))
            s.close()
        tol = 1e-11 if dtype == np.float64 else 2e-5
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=tol)
        for x, y in zip(outs[0][1] + outs[0][2], outs[1][1] + outs[1][2]):
            assert np.linalg.norm(np.asarray(x, np.float64) - y) <= 10 * tol * np.linalg.norm(y)
        assert outs[0][3] < outs[1][3]  # one tile copy per baseline instead of one per (slice, baseline)


@pytest.mark.parametrize("nfreqs,nvecs", [(300, [4, 16, 17, 33, 48, 49, 64, 80, 96, 112, 113, 128, 144, 160, 176, 192, 193, 208, 224]),
                                          (200, [3, 20, 40, 56, 57, 100, 112, 113, 129, 145, 160])])
def test_slices_share_tiles_band_shapes_and_the_one_pass_regulariser(nfreqs, nvecs):
    """(a) 300 channels pad to 384 = six 16-channel strips per wave: not a multiple of four, so the blocks of at most 48 vectors run
    fused_multi_mfma_kernel's two-deep ring instead of the four-deep one (every other test here, and HERA-350's 1024 channels,
    take the four-deep form); every class of both precisions, members 8 + 3 (16 MFMA columns in float32 and float64).
    (b) Every head on the matrix-core kernel (no block wider than 224 vectors; float64: 160): the regularised gradient pass is ONE
    pass with two adjoint sets (REG == 2: S, q1, the second coefficient gradients, folded by the combine kernels / the one-launch
    tail) -- float32 in both cases, float64 in the second; the first case in float64 keeps the two passes.
    Loss and gradients against the C restatement of the batched problem, with and without the regulariser, and a regularised Adam
    trajectory against the same problem without the alias table (every baseline streams its own tiles through fused_basis_kernel)."""
    from calamity_amd import distributed as D
    from calamity_amd.solver import HipFitSolver

    nsl = 11
    base, _ = random_problem(nvecs, [1] * len(nvecs), nants=7, nfreqs=nfreqs, seed=90)
    parts = []
    for t in range(nsl):
        p, st = random_problem(nvecs, [1] * len(nvecs), nants=7, nfreqs=nfreqs, seed=91 + t)
        p.basis, p.grp_basis = base.basis, base.grp_basis
        p.bl_ant0, p.bl_ant1, p.bl_rowblk = base.bl_ant0, base.bl_ant1, base.bl_rowblk
        p.wgts = p.wgts / nsl
        parts.append((p, st))
    prob, start = D.batch_time_slices(parts)
    assert prob.bl_alias is not None
    prob.sky_r, prob.sky_i = np.concatenate([p.sky_r for p, _ in parts]), np.concatenate([p.sky_i for p, _ in parts])
    check(prob, start, layouts=("stream",))
    plain = D.batch_time_slices(parts)[0]
    plain.bl_alias = None
    pri = float(np.sum(prob.sky_r * prob.wgts)), float(np.sum(prob.sky_i * prob.wgts))
    for dtype in (np.float64, np.float32):
        outs = []
        for pr in (prob, plain):
            s = HipFitSolver(dtype=dtype)
            s.set_problem(pr, layout="stream")
            s.set_params(start["g_r"], start["g_i"], start["c_r"], start["c_i"])
            s.set_regularization("sum", *pri)
            s.set_optimizer("Adam", learning_rate=1e-2)
            losses, _, _ = s.run(6, record=True, tol=0.0)
            outs.append((losses, s.get_params()))
            s.close()
        tol = 1e-11 if dtype == np.float64 else 2e-5
        np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=tol)
        for x, y in zip(outs[0][1], outs[1][1]):
            assert np.linalg.norm(np.asarray(x, np.float64) - y) <= 10 * tol * np.linalg.norm(y)


def test_slices_that_share_only_some_tiles():
    """A solver that holds ordinary baselines beside sets that share tiles: the alias table of a batched problem with every
    third baseline of the later slices given its own tile copy again (bl_alias = -1), and one whole slice un-aliased.  The
    ordinary items are launched on fused_basis_kernel, the covered ones on the multi-slice kernels (the host orders covered
    items last); loss and gradients against the C restatement, which knows nothing of the table."""
    from calamity_amd import distributed as D

    nvecs = [3, 40, 56, 90, 112, 150, 224, 225, 17, 64]
    base, _ = random_problem(nvecs, [1] * len(nvecs), nants=9, nfreqs=200, seed=70)
    parts = []
    for t in range(6):
        p, st = random_problem(nvecs, [1] * len(nvecs), nants=9, nfreqs=200, seed=71 + t)
        p.basis, p.grp_basis = base.basis, base.grp_basis
        p.bl_ant0, p.bl_ant1, p.bl_rowblk = base.bl_ant0, base.bl_ant1, base.bl_rowblk
        p.wgts = p.wgts / 6
        parts.append((p, st))
    prob, start = D.batch_time_slices(parts)
    alias = prob.bl_alias.copy()
    alias[base.nbls + 1::3] = -1                       # every third baseline of the later slices
    alias[3 * base.nbls: 4 * base.nbls] = -1           # all of slice 3
    assert np.any(alias >= 0) and np.any(alias[base.nbls:] < 0)
    prob.bl_alias = alias
    prob.sky_r, prob.sky_i = np.concatenate([p.sky_r for p, _ in parts]), np.concatenate([p.sky_i for p, _ in parts])
    check(prob, start, layouts=("stream",))


@pytest.mark.parametrize("seed", range(8))
def test_randomly_drawn_slices_that_share_tiles(seed):
    """Seeded random batches of time slices: 2 to 10 slices, 1 to 224 (and a few more) vectors per block, bands whose padded rows
    give a wave 2, 4, 6, 8 or 16 strips (the deep tile-register rings need a multiple of 4 and fall back otherwise), a random part
    of the alias table switched off -- float32 loss and gradients of the matrix-core multi-slice kernel against the C restatement,
    float64 for the vector-ALU form."""
    from calamity_amd import distributed as D

    rng = np.random.default_rng(3000 + seed)
    nslices = int(rng.integers(2, 11))
    nfreqs = int(rng.choice([65, 128, 200, 256, 300, 384, 500, 1000]))
    ngrps = int(rng.integers(1, 7))
    nvecs = [int(rng.integers(1, min(nfreqs, 240) + 1)) for _ in range(ngrps)]
    nants = int(rng.integers(3, 10))
    base, _ = random_problem(nvecs, [1] * ngrps, nants=nants, nfreqs=nfreqs, seed=3100 + seed)
    parts = []
    for t in range(nslices):
        p, st = random_problem(nvecs, [1] * ngrps, nants=nants, nfreqs=nfreqs, seed=3200 + 16 * seed + t)
        p.basis, p.grp_basis = base.basis, base.grp_basis
        p.bl_ant0, p.bl_ant1, p.bl_rowblk = base.bl_ant0, base.bl_ant1, base.bl_rowblk
        p.wgts = p.wgts / nslices
        parts.append((p, st))
    prob, start = D.batch_time_slices(parts)
    if seed % 2:
        alias = prob.bl_alias.copy()
        off = rng.random(alias.size) < 0.2
        alias[off] = -1
        prob.bl_alias = alias
    prob.sky_r, prob.sky_i = np.concatenate([p.sky_r for p, _ in parts]), np.concatenate([p.sky_i for p, _ in parts])
    check(prob, start, layouts=("stream",), regs=(False,))


@pytest.mark.parametrize("nfreqs", [24, 64, 65])
def test_slices_that_share_tiles_of_a_narrow_band(nfreqs):
    """Bands of at most 64 channels are padded to 8 ... 64 channels, not to a multiple of 128: the matrix-core multi-slice kernel
    (whose waves take 16-channel strips in pairs) must leave them to fused_multi_kernel; 65 channels are the first it takes."""
    from calamity_amd import distributed as D

    nvecs = [3, 20, min(nfreqs, 50)]
    base, _ = random_problem(nvecs, [1] * len(nvecs), nants=6, nfreqs=nfreqs, seed=80)
    parts = []
    for t in range(5):
        p, st = random_problem(nvecs, [1] * len(nvecs), nants=6, nfreqs=nfreqs, seed=81 + t)
        p.basis, p.grp_basis = base.basis, base.grp_basis
        p.bl_ant0, p.bl_ant1, p.bl_rowblk = base.bl_ant0, base.bl_ant1, base.bl_rowblk
        p.wgts = p.wgts / 5
        parts.append((p, st))
    prob, start = D.batch_time_slices(parts)
    prob.sky_r, prob.sky_i = np.concatenate([p.sky_r for p, _ in parts]), np.concatenate([p.sky_i for p, _ in parts])
    check(prob, start, layouts=("stream",), regs=(False,))


@pytest.mark.parametrize("seed", range(10))
def test_randomly_drawn_problems(seed):
    """Seeded random problems over the corners the fixed cases above do not name: 2 antennas, 1 to 5 channels, channel counts
    around every padding boundary, single-vector blocks, blocks as wide as the band, groups of 1-4 baselines in any mix --
    loss and every gradient of both layouts and precisions, with and without the regulariser, against the C restatement."""
    rng = np.random.default_rng(1000 + seed)
    nants = int(rng.integers(2, 11))
    nfreqs = int(rng.choice([1, 2, 5, 8, 9, 31, 64, 65, 127, 128, 129, 200, 257]))
    ngrps = int(rng.integers(1, 9))
    nvecs = [int(rng.integers(1, min(max(nfreqs, 1), 260) + 1)) for _ in range(ngrps)]
    if seed % 3 == 0:
        nvecs[0] = 1
    bls = [int(rng.integers(1, 5)) for _ in range(ngrps)]
    p, start = random_problem(nvecs, bls, nants=nants, nfreqs=nfreqs, seed=2000 + seed, rowblocks=bool(seed % 2))
    check(p, start)
