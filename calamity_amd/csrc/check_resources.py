#!/usr/bin/env python3
"""Build-time guard (called by the Makefile with hipcc's -Rpass-analysis=kernel-resource-usage remarks on stdin).

The dense kernels consume their LDS-DMA operand ring behind COUNTED `s_waitcnt vmcnt(N)`: every vector-memory operation of
a wave is in that count, and register spills are vector-memory operations (scratch).  A spilling build would therefore not
just be slower, it would read ring slots before they have landed.  So: no scratch, no spills, in any kernel whose name
contains `fused_dense`; and two waves per SIMD where the kernels are written for two.
`fused_multi_mfma_kernel<float>` sits at 245 of the 256 registers two waves per SIMD leave it (the double instance has a
CU to itself): no scratch and no spilled VGPRs there either (a few SGPR spills into lanes of a VGPR are tolerated: they sit outside its job loop)."""
import re
import sys

cur, bad, seen = None, [], 0
for line in sys.stdin:
    m = re.search(r"remark:\s+Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        seen += "fused_dense" in cur
        continue
    if cur is not None and "fused_multi_mfma" in cur:
        m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
        if m and m.group(1).startswith("Occupancy"):
            inst = re.search(r"fused_multi_mfma_kernelI([fd])Li(\d)ELi(\d)E", cur)  # <T, MODE, REG>
            one_wg = inst is None or inst.group(1) == "d" or inst.group(3) == "2"  # (double, and the one-pass regularised form: one workgroup per CU by design)
            if int(m.group(2)) < 2 and not one_wg:
                bad.append(f"{cur}: occupancy {m.group(2)} waves/SIMD (< 2)")
        elif m and int(m.group(2)) != 0:
            bad.append(f"{cur}: {m.group(1)} = {m.group(2)}")
        continue
    if cur is None or "fused_dense" not in cur:
        continue
    m = re.search(r"remark:\s+(ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", line)
    if not m:
        continue
    key, val = m.group(1), int(m.group(2))
    if key.startswith("Occupancy"):
        if val < 2 and "fused_dense_split" not in cur:  # (the split-bf16 kernel runs one workgroup per CU by design: split_kernels.hpp)
            bad.append(f"{cur}: occupancy {val} waves/SIMD (< 2)")
    elif val != 0 and not (key.startswith("SGPRs Spill") and "fused_dense_split" in cur):
        # (SGPR spills of the split-bf16 kernel go into lanes of a VGPR -- it has 512 -- not to scratch: ScratchSize stays checked)
        bad.append(f"{cur}: {key} = {val}")
if seen == 0:
    bad.append("no fused_dense kernel found in the compiler's resource remarks")
if bad:
    sys.stderr.write("check_resources: the dense kernels must not spill (scratch traffic breaks their counted waits):\n  " + "\n  ".join(bad) + "\n")
    sys.exit(1)
print(f"check_resources: {seen} dense kernels, no scratch, no spills")
