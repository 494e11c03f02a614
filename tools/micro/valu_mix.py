#!/usr/bin/env python3
"""(GPU box) Calibration of the vector-issue roofline (VERDICT r5, item 4 iii): what rate of wave64 vector instructions does an MI355X sustain
for the INSTRUCTION MIX of the path kernels?

bench.py's `roofline.frac` counts every vector instruction at the 2 cycles a plain f32 op occupies a SIMD-32; round 5 priced the mix with a cost
model (2 / 4 / 8 cycles per class) whose constants were loose by 10 %. This tool measures instead: from the committed PMC summary of a kernel
(profiles/rNN_pmc_summary*.json: SQ_INSTS_VALU and its per-class counters) and the static opcode histogram of the kernel's ISA (how a class splits
into opcodes: --asm <listing>, default = the table below, taken from the plain pt kernel) it generates a straight-line kernel with the same mix over
sixteen independent dependency chains, runs it at the path kernel's geometry (1024-thread workgroups = 4 waves per SIMD, one per CU, all 256 CUs) and
prints the vector instructions per second it reaches -- the peak the path kernel's own rate is divided by (`roofline.frac_of_mix_peak`).

    python3 tools/micro/valu_mix.py profiles/r06_pmc_summary.json [profiles/r06_pmc_summary_ptdl.json ...] > profiles/r06_valu_mix_peak.json
"""
import json
import os
import re
import subprocess
import sys
import tempfile

# how the PMC classes split into opcodes: static counts of the plain pt kernel's ISA (round 6), per class
DEFAULT_SPLIT = {
    "add_f32": {"v_add_f32": 523, "v_sub_f32": 297},
    "mul_f32": {"v_mul_f32": 1080},
    "fma_f32": {"v_fma_f32": 471, "v_fmac_f32": 326},
    "trans": {"v_rcp_f32": 129, "v_sqrt_f32": 52},
    "int32": {"v_xor_b32": 451, "v_add_u32": 349, "v_or_b32": 98, "v_and_b32": 83, "v_lshrrev_b32": 78, "v_lshlrev_b32": 55, "v_alignbit_b32": 91,
              "v_mul_lo_u32": 34, "v_lshl_add_u32": 30, "v_bfe_u32": 23},
    "int64": {"v_lshrrev_b64": 112, "v_mad_u64_u32": 107, "v_lshl_add_u64": 91, "v_lshlrev_b64": 61},
    "cvt": {"v_cvt_f32_u32": 1},
    "f64": {"v_mul_f64": 35, "v_fma_f64": 15, "v_add_f64": 10},
    # what no class counter counts: selects, moves, compares, min / max, the division helpers. Its DYNAMIC composition is not measurable (no counter, no PC
    # sampling on this pool), so the class is generated twice and the peak is reported as a bracket:
    #   "other"       the whole kernel's static histogram (over-weights cold shading code: IEEE division helpers)
    #   "other_hot"   what the hottest loop -- the node visit, a third of all executed vector instructions -- is made of besides f32 math:
    #                 21 selects, 4 compares, 8 min / max + 8 three-operand min3 / max3, a few moves (csrc/mi_kernels.h: node_visit)
    "other": {"v_cndmask_b32": 607, "v_mov_b32": 554, "v_cmp_lt_f32": 520, "v_cmp_eq_u32": 250, "v_div_scale_f32": 163, "v_div_fixup_f32": 123,
              "v_div_fmas_f32": 82, "v_max_f32": 71, "v_readlane_b32": 64},
    "other_hot": {"v_cndmask_b32": 21, "v_cmp_lt_f32": 4, "v_max_f32": 8, "v_max3_f32": 8, "v_mov_b32": 6},
}

# one instance of an opcode on chain k: F = float accumulators f0..f15, I = ints, D = doubles (4), S = scalar pairs for compare results
TEMPLATES = {
    "v_add_f32": "v_add_f32 {F}, {F}, {c1}", "v_sub_f32": "v_sub_f32 {F}, {F}, {c1}", "v_mul_f32": "v_mul_f32 {F}, {F}, {c0}",
    "v_fma_f32": "v_fma_f32 {F}, {F}, {c0}, {c1}", "v_fmac_f32": "v_fmac_f32 {F}, {c0}, {c1}",
    "v_rcp_f32": "v_rcp_f32 {F}, {F}", "v_sqrt_f32": "v_sqrt_f32 {F}, {F}",
    "v_xor_b32": "v_xor_b32 {I}, {I}, {ci}", "v_add_u32": "v_add_u32 {I}, {I}, {ci}", "v_or_b32": "v_or_b32 {I}, {I}, {ci}", "v_and_b32": "v_and_b32 {I}, {I}, {ci}",
    "v_lshrrev_b32": "v_lshrrev_b32 {I}, 1, {I}", "v_lshlrev_b32": "v_lshlrev_b32 {I}, 1, {I}", "v_alignbit_b32": "v_alignbit_b32 {I}, {I}, {ci}, 7",
    "v_mul_lo_u32": "v_mul_lo_u32 {I}, {I}, {ci}", "v_lshl_add_u32": "v_lshl_add_u32 {I}, {I}, 2, {ci}", "v_bfe_u32": "v_bfe_u32 {I}, {I}, 3, 9",
    "v_lshrrev_b64": "v_lshrrev_b64 {D}, 1, {D}", "v_mad_u64_u32": "v_mad_u64_u32 {D}, s[36:37], {I}, {ci}, {D}", "v_lshl_add_u64": "v_lshl_add_u64 {D}, {D}, 1, {D}",
    "v_lshlrev_b64": "v_lshlrev_b64 {D}, 1, {D}",
    "v_cvt_f32_u32": "v_cvt_f32_u32 {F}, {I}",
    "v_mul_f64": "v_mul_f64 {D}, {D}, {cd}", "v_fma_f64": "v_fma_f64 {D}, {D}, {cd}, {cd}", "v_add_f64": "v_add_f64 {D}, {D}, {cd}",
    "v_cndmask_b32": "v_cndmask_b32 {F}, {F}, {c0}, {S}", "v_mov_b32": "v_mov_b32 {F}, {c0}", "v_cmp_lt_f32": "v_cmp_lt_f32 {S}, {F}, {c1}",
    "v_cmp_eq_u32": "v_cmp_eq_u32 {S}, {I}, {ci}", "v_div_scale_f32": "v_div_scale_f32 {F}, {S}, {F}, {c0}, {c1}", "v_div_fixup_f32": "v_div_fixup_f32 {F}, {F}, {c0}, {c1}",
    "v_div_fmas_f32": "v_div_fmas_f32 {F}, {F}, {c0}, {c1}", "v_max_f32": "v_max_f32 {F}, {F}, {c1}", "v_max3_f32": "v_max3_f32 {F}, {F}, {c0}, {c1}", "v_readlane_b32": "v_readlane_b32 s38, {I}, 3",
}
CHAINS = 16
BLOCK = 480          # vector instructions of the generated block


def classes_of(summary):
    g = lambda k: float(summary.get(k, 0.0))
    total = g("SQ_INSTS_VALU")
    c = {"add_f32": g("SQ_INSTS_VALU_ADD_F32"), "mul_f32": g("SQ_INSTS_VALU_MUL_F32"), "fma_f32": g("SQ_INSTS_VALU_FMA_F32"), "trans": g("SQ_INSTS_VALU_TRANS_F32"),
         "int32": g("SQ_INSTS_VALU_INT32"), "int64": g("SQ_INSTS_VALU_INT64"), "cvt": g("SQ_INSTS_VALU_CVT"),
         "f64": g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")}
    c["other"] = max(0.0, total - sum(c.values()))
    return {k: v / total for k, v in c.items()}


def split_from_asm(path):
    """the kernel's own static opcode histogram, sorted into the classes of DEFAULT_SPLIT by opcode (unknown opcodes go to `other` as v_cndmask_b32)"""
    known = {op: cls for cls, ops in DEFAULT_SPLIT.items() for op in ops}
    out = {cls: {} for cls in DEFAULT_SPLIT}
    for line in open(path):
        m = re.match(r"\s+(v_[a-z0-9_]+)", line)
        if not m:
            continue
        op = re.sub(r"_(e32|e64|sdwa|dpp)$", "", m.group(1))
        if op in known:
            out[known[op]][op] = out[known[op]].get(op, 0) + 1
    return {cls: (ops if ops else DEFAULT_SPLIT[cls]) for cls, ops in out.items()}


def generate(frac, split, hot=False):
    """BLOCK instructions with the classes' shares, each class split by its opcodes, interleaved so that no class clusters"""
    want = []
    for cls, f in frac.items():
        ops = split[cls if not (cls == "other" and hot) else "other_hot"]
        tot = float(sum(ops.values()))
        for op, n in ops.items():
            want.append((op, f * n / tot * BLOCK))
    counts = {op: int(round(x)) for op, x in want}
    seq = []
    # error-diffusion interleave: at every slot emit the opcode that is furthest behind its share
    done = {op: 0 for op in counts}
    n_total = sum(counts.values())
    for i in range(n_total):
        op = max(counts, key=lambda o: (counts[o] * (i + 1) / n_total - done[o]) if counts[o] else -1e9)
        done[op] += 1
        seq.append(op)
    lines = []
    for i, op in enumerate(seq):
        k = i % CHAINS
        lines.append(TEMPLATES[op].format(F=f"%{k}", I=f"%{CHAINS + k % 8}", D=f"%{CHAINS + 8 + k % 4}", S=f"s[{20 + 2 * (k % 8)}:{21 + 2 * (k % 8)}]",
                                          c0=f"%{CHAINS + 12}", c1=f"%{CHAINS + 13}", ci=f"%{CHAINS + 14}", cd=f"%{CHAINS + 15}"))
    return lines, counts


SRC = r"""
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(1024) k_mix(float *out, int iters, unsigned long long *clk)
{
  extern __shared__ float lds_pad[];          /* 100 KB per workgroup: ONE workgroup per CU, i.e. 4 waves per SIMD on all 256 CUs -- the path kernel's geometry */
  if(iters < 0) lds_pad[threadIdx.x] = 0.0f;
  const unsigned long long t0 = clock64(), w0 = wall_clock64();
  float f[16]; unsigned int n[8]; double d[4];
  for(int k=0;k<16;k++) f[k] = 1.0f + threadIdx.x*1e-3f + k;
  for(int k=0;k<8;k++) n[k] = threadIdx.x*2654435761u + k;
  for(int k=0;k<4;k++) d[k] = 1.0 + threadIdx.x*1e-3 + k;
  float c0 = 0.9990234375f, c1 = 1e-3f; unsigned int ci = 0x9e3779b9u; double cd = 0.99951171875;
  for(int i=0;i<iters;i++)
  {
    asm volatile(
%s
      : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]), "+v"(f[8]), "+v"(f[9]), "+v"(f[10]), "+v"(f[11]), "+v"(f[12]), "+v"(f[13]), "+v"(f[14]), "+v"(f[15]),
        "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3])
      : "v"(c0), "v"(c1), "v"(ci), "v"(cd)
      : "vcc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "s30", "s31", "s32", "s33", "s34", "s35", "s36", "s37", "s38");
  }
  float s = 0; for(int k=0;k<16;k++) s += f[k]; for(int k=0;k<8;k++) s += (float)n[k]; for(int k=0;k<4;k++) s += (float)d[k];
  out[blockIdx.x*blockDim.x + threadIdx.x] = s;
  if(clk && threadIdx.x == 0) { clk[2*blockIdx.x] = clock64() - t0; clk[2*blockIdx.x + 1] = wall_clock64() - w0; }     /* shader clock ticks, 100 MHz ticks */
}
int main()
{
  float *dout; if(hipMalloc(&dout, 256*1024*sizeof(float)) != hipSuccess) return 1;
  unsigned long long *dclk, hclk[512]; if(hipMalloc(&dclk, sizeof(hclk)) != hipSuccess) return 1;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, block = %d;
  if(hipFuncSetAttribute((const void *)k_mix, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024) != hipSuccess) return 2;
  k_mix<<<256, 1024, 100*1024>>>(dout, 50, nullptr);
  double best = 1e30;
  for(int r=0;r<5;r++)
  {
    hipEventRecord(e0); k_mix<<<256, 1024, 100*1024>>>(dout, iters, dclk); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if(ms < best) best = ms;
  }
  const double waves = 256.0*16.0, instr = waves*(double)iters*block;
  hipMemcpy(hclk, dclk, sizeof(hclk), hipMemcpyDeviceToHost);
  double ticks = 0, wall = 0; for(int b=0;b<256;b++) { ticks += (double)hclk[2*b]; wall += (double)hclk[2*b + 1]; }
  printf("{\"ms\": %%.4f, \"wave_instructions\": %%.0f, \"ginstr_per_s\": %%.2f, \"cycles_per_instruction_per_simd_at_2p4GHz\": %%.3f, \"clock64_over_wall_clock64\": %%.3f}\n",
         best, instr, instr/(best*1e-3)/1e9, best*1e-3*2.4e9/(16.0/4.0*(double)iters*block), ticks/wall);
  return 0;
}
"""


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    asm = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--asm=")), None)
    split = split_from_asm(asm) if asm else DEFAULT_SPLIT
    out = {"what": "vector instructions per second a generated kernel with the path kernel's instruction mix sustains: 256 workgroups of 1024 threads (4 waves per SIMD), "
                   "sixteen independent chains, no memory access (tools/micro/valu_mix.py)", "opcode_split": "kernel ISA" if asm else "default table (plain pt kernel, round 6)"}
    for path in args:
        summary = json.load(open(path))
        frac = classes_of(summary)
        res = {"kernel": summary.get("kernel"), "class_shares": frac}
        for tag, hot in (("uncounted_class_as_whole_kernel", False), ("uncounted_class_as_node_visit", True)):
            lines, counts = generate(frac, split, hot)
            body = "\n".join('      "%s\\n"' % l for l in lines)
            with tempfile.TemporaryDirectory() as td:
                src = os.path.join(td, "valu_mix_gen.hip")
                open(src, "w").write(SRC % (body, len(lines)))
                exe = os.path.join(td, "valu_mix_gen")
                subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", src, "-o", exe], stderr=subprocess.DEVNULL)
                r = {"built": True} if "--build-only" in sys.argv else json.loads(subprocess.check_output([exe], text=True).strip().splitlines()[-1])
            r.update({"block_instructions": len(lines), "block_opcodes": counts})
            res[tag] = r
        if "--build-only" not in sys.argv:
            res["peak_ginstr_per_s"] = sorted(res[t]["ginstr_per_s"] for t in ("uncounted_class_as_whole_kernel", "uncounted_class_as_node_visit"))
        out[os.path.basename(path)] = res
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
