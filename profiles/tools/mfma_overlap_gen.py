#!/usr/bin/env python3
"""Generator of profiles/tools/mfma_overlap_probe.cpp (profiling tool, not product; r05).

Question: how much vector-ALU work does a gfx950 SIMD issue UNDER a running v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 operands, 8 passes =
32 cycles)?  The level-0 Gram kernel of the discrete kinds (fw_mi.hip: mi_level0_mfma_kernel) issues 8 matrix instructions and ~45
expansion instructions per wavefront and 64-sample word with two wavefronts per SIMD and takes ~966 cycles per word and SIMD where the
matrix pipe alone needs 512.  Every kernel below is ONE asm block with fixed physical registers, so the instruction stream is exactly
what is written here: per word 8 matrix instructions on 8 independent accumulators, and between two of them NV vector instructions that
are (dep = 1) the expansion sequence writing the operand registers of the NEXT word or (dep = 0) the same instructions on registers no
matrix instruction reads; lds = 1 adds the six ds_read_b32 of a word.  Two words per loop trip (the operand buffers alternate).

usage: python profiles/tools/mfma_overlap_gen.py > profiles/tools/mfma_overlap_probe.cpp
       hipcc --offload-arch=gfx950 -O2 profiles/tools/mfma_overlap_probe.cpp -o profiles/tools/mfma_overlap_probe.bin
"""
import sys

M = "s30"          # 0x22222222
ACC = lambda k: "v[%d:%d]" % (16 * k, 16 * k + 15)
XOP = lambda buf, i: 128 + 24 * buf + 4 * i       # i = 0..3: (block a, plane px) = (i >> 1, i & 1)
YOP = lambda buf, j: 128 + 24 * buf + 16 + 4 * j  # j = 0..1
RAW = lambda buf, i: 176 + 6 * buf + i            # six raw words per buffer
SCR = 192                                         # scratch destinations of the independent form
SCALE = "v190"


def expand(dst, raw):
    """seven instructions: four operand registers from one 32-sample word"""
    return ["v_lshlrev_b32 v%d, 1, v%d" % (dst, raw), "v_and_b32 v%d, %s, v%d" % (dst, M, dst), "v_and_b32 v%d, %s, v%d" % (dst + 1, M, raw),
            "v_lshrrev_b32 v%d, 1, v%d" % (dst + 2, raw), "v_and_b32 v%d, %s, v%d" % (dst + 2, M, dst + 2),
            "v_lshrrev_b32 v%d, 2, v%d" % (dst + 3, raw), "v_and_b32 v%d, %s, v%d" % (dst + 3, M, dst + 3)]


def word(buf, nv, dep, lds, valu_first=False):
    """instruction list of one word: consumes operand buffer `buf`, produces buffer 1 - buf"""
    other = 1 - buf
    valu = []
    for i in range(4):
        valu += expand(XOP(other, i) if dep else SCR + 4 * (i & 1), RAW(other, i))
    for j in range(2):
        valu += expand(YOP(other, j) if dep else SCR + 8 + 4 * (j & 1), RAW(other, 4 + j))
    valu = valu[:8 * nv] if nv * 8 <= len(valu) else valu + valu[:8 * nv - len(valu)]
    out = []
    if lds:
        for i in range(6):
            out.append("ds_read_b32 v%d, v191 offset:%d" % (RAW(other, i), 64 * 4 * i))
        out.append("s_waitcnt lgkmcnt(0)")
    m = 0
    for a in range(2):
        for px in range(2):
            for py in range(2):
                k = a * 4 + px * 2 + py
                out.append("v_mfma_scale_f32_32x32x64_f8f6f4 %s, v[%d:%d], v[%d:%d], %s, %s, %s op_sel_hi:[0,0,0] cbsz:4 blgp:4" %
                           (ACC(k), XOP(buf, a * 2 + px), XOP(buf, a * 2 + px) + 3, YOP(buf, py), YOP(buf, py) + 3, ACC(k), SCALE, SCALE))
                out += valu[m * nv:(m + 1) * nv]
                m += 1
    return out


def kernel(name, nv, dep, lds, mfma=True, rnd=False):
    body = word(0, nv, dep, lds) + word(1, nv, dep, lds)
    if not mfma:
        body = [l for l in body if not l.startswith("v_mfma")]
    init = ["s_mov_b32 s29, 0x85ebca6b", "s_mov_b32 %s, 0x22222222" % M, "v_mov_b32 %s, 0x7f7f7f7f" % SCALE, "v_lshlrev_b32 v191, 2, %1"]
    init += ["v_mov_b32 v%d, 0" % r for r in range(0, 128)]
    if rnd:  # dense pseudo-random bits in the raw words and the operand registers (what real bit planes look like to the matrix pipe)
        init += ["v_mul_lo_u32 v%d, %%1, s29" % r for r in range(128, 190)]
        init += ["v_xor_b32 v%d, 0x%08x, v%d" % (r, (0x9e3779b9 * (r + 1)) & 0xffffffff, r) for r in range(128, 190)]
        init += ["v_and_b32 v%d, s30, v%d" % (r, r) for r in range(128, 176)]  # operand registers: valid E2M1 codes 0 / 1.0 only
    else:
        init += ["v_mov_b32 v%d, %%1" % r for r in range(128, 190)]
    init += ["v_mov_b32 v%d, 0" % r for r in range(192, 208)]
    loop = ["s_mov_b32 s31, %2", "1:"] + body + ["s_sub_u32 s31, s31, 1", "s_cmp_lg_u32 s31, 0", "s_cbranch_scc1 1b", "s_nop 15", "s_nop 15"]
    fin = ["v_add_f32 v0, v0, v16", "v_add_f32 v0, v0, v32", "v_add_f32 v0, v0, v48", "v_add_f32 v0, v0, v64", "v_add_f32 v0, v0, v80",
           "v_add_f32 v0, v0, v96", "v_add_f32 v0, v0, v112", "v_add_u32 v0, v0, v192", "v_add_u32 v0, v0, v200", "v_mov_b32 %0, v0"]
    clob = ", ".join('"v%d"' % r for r in range(0, 208)) + ', "s29", "s30", "s31", "scc", "memory"'
    asm = "\\n\"\n        \"".join(init + loop + fin)
    n_valu = sum(1 for l in body if l.startswith(("v_and", "v_lsh")))
    n_mfma = sum(1 for l in body if l.startswith("v_mfma"))
    return n_valu, n_mfma, """
__global__ __launch_bounds__(512) void %s(float *out, int iters, unsigned long long *cyc)
{
    __shared__ unsigned s_w[2048];
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) s_w[i] = %s;
    __syncthreads();
    float r;
    const unsigned lane = threadIdx.x & 63u;
    const unsigned long long t0 = __builtin_readcyclecounter();
    asm volatile("%s\\n"
                 : "=v"(r)
                 : "v"(lane), "s"(iters)
                 : %s);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (r == 12345.5f) out[0] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
""" % (name, "0x9e3779b9u * (i + 1)" if rnd else "0u", asm, clob)


VARIANTS = [  # name, VALU per matrix instruction, dependent?, ds_reads?, matrix instructions present?
    ("k_mfma_only", 0, 0, 0, True),
    ("k_v2_indep", 2, 0, 0, True),
    ("k_v3_indep", 3, 0, 0, True),
    ("k_v4_indep", 4, 0, 0, True),
    ("k_v5_indep", 5, 0, 0, True),
    ("k_v6_indep", 6, 0, 0, True),
    ("k_v3_dep", 3, 1, 0, True),
    ("k_v5_dep", 5, 1, 0, True),
    ("k_v6_dep", 6, 1, 0, True),
    ("k_v5_dep_lds", 5, 1, 1, True),
    ("k_v3_dep_lds", 3, 1, 1, True),
    ("k_valu_only_v5", 5, 1, 0, False),
    ("k_valu_only_v3", 3, 1, 0, False),
    ("k_mfma_only_rnd", 0, 0, 0, True, True),
    ("k_v5_dep_rnd", 5, 1, 0, True, True),
    ("k_v3_dep_rnd", 3, 1, 0, True, True),
    ("k_v5_dep_lds_rnd", 5, 1, 1, True, True),
    ("k_v3_dep_lds_rnd", 3, 1, 1, True, True),
]

print("// GENERATED by profiles/tools/mfma_overlap_gen.py -- do not edit.  Profiling tool, not product.")
print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <vector>\n#include <algorithm>")
meta = []
for v in VARIANTS:
    name, nv, dep, lds, mf = v[:5]
    nval, nmf, src = kernel(name, nv, dep, lds, mf, len(v) > 5 and v[5])
    meta.append((name, nval, nmf))
    print(src)
print("""
typedef void (*kfn)(float *, int, unsigned long long *);
static void run(const char *name, kfn k, int threads, int n_valu, int n_mfma)
{
    const int iters = 4000, grid = 256;
    float *d;
    unsigned long long *c;
    hipMalloc(&d, 4);
    hipMalloc(&c, 8 * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, iters, c);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), 0, 0, d, iters, c);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), c, 8 * grid, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const double words = 2.0 * iters;  // per wavefront
    const int wps = threads / 256;     // wavefronts per SIMD
    printf("%-16s %d wave(s)/SIMD: %7.3f ms = %7.1f ns per word and SIMD (%6.1f cycles at the nominal %.2f GHz; counter: %6.1f per word, median workgroup); "
           "per word and SIMD: %d matrix + %d vector instructions; matrix rate %.2f PF\\n",
           name, wps, ms, 1e6 * ms / words, 1e6 * ms / words * clk_khz * 1e-6, clk_khz * 1e-6, (double)h[grid / 2] / words, wps * n_mfma / 2, wps * n_valu / 2,
           n_mfma ? 2.0 * 65536.0 * (n_mfma / 2) * words * (threads / 64) * grid / (ms * 1e-3) * 1e-15 : 0.0);
    hipFree(d);
    hipFree(c);
}
int main()
{
    for (int threads = 256; threads <= 512; threads += 256) {""")
for name, nval, nmf in meta:
    print('        run("%s", %s, threads, %d, %d);' % (name, name, nval, nmf))
print("""    }
    return 0;
}""")
