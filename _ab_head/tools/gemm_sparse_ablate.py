#!/usr/bin/env python3
"""Where does the dropout-aware projection (k_gemm_nn_sparse<4, true>) lose its time?  Runs ON THE GPU BOX: patches a scratch
copy of csrc/gemm.hip (the snapshot there is thrown away), rebuilds, times the kernel at the config-F shape (B 64, T 1900,
F 1000, H 500, p = 0.5).  Variants that drop work compute wrong numbers - only their TIME is of interest."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-gesture-recognition-with-lstms-and-ctc_amd")
SRC = os.path.join(PKG, "csrc", "gemm.hip")
orig = open(SRC).read()

MFMA0 = "acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bb, acc[g][0], 0, 0, 0);"
MFMA1 = "acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bb, acc[g][1], 0, 0, 0);"
AFETCH = "const float4 v0 = *reinterpret_cast<const float4*>(xp), v1 = *reinterpret_cast<const float4*>(xp + 64);"
BFETCH = "r.w[j] = Wg[(size_t)Ls[q] * H];"
VARIANTS = {
    "baseline": [],
    "no MFMA (LDS reads kept alive by one FMA each)": [(MFMA0, "acc[g][0][0] += a0 * bb;"), (MFMA1, "acc[g][1][0] += a1 * bb;")],
    "no global A loads": [(AFETCH, "const float4 v0 = make_float4(1.f, 2.f, 3.f, (float)st), v1 = v0; (void)xp;")],
    "no global B gather": [(BFETCH, "r.w[j] = (float)q;")],
    "no global loads at all": [(AFETCH, "const float4 v0 = make_float4(1.f, 2.f, 3.f, (float)st), v1 = v0; (void)xp;"), (BFETCH, "r.w[j] = (float)q;")],
    "no MFMA, no global loads (LDS + barriers + epilogue)": [(MFMA0, "acc[g][0][0] += a0 * bb;"), (MFMA1, "acc[g][1][0] += a1 * bb;"),
                                                             (AFETCH, "const float4 v0 = make_float4(1.f, 2.f, 3.f, (float)st), v1 = v0; (void)xp;"), (BFETCH, "r.w[j] = (float)q;")],
}
try:
    for name, subs in VARIANTS.items():
        s = orig
        for a, b in subs:
            assert s.count(a) == 1, a
            s = s.replace(a, b)
        open(SRC, "w").write(s)
        r = subprocess.run([sys.executable, os.path.join(PKG, "_build.py"), "--force"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            print(name, ": build failed\n", r.stdout.decode()[-2000:])
            continue
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_bench.py"), "--what", "gemm1"], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT).stdout.decode()
        line = [l for l in out.split("\n") if "transposed copy" in l]
        print("%-55s %s" % (name, line[0].strip() if line else out[-400:]), flush=True)
finally:
    open(SRC, "w").write(orig)
