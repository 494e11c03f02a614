#!/usr/bin/env python3
"""development helper (build container, nothing runs on a GPU): the vector instructions ONE execution of each block of the hot path needs,
read from the code of tools/micro/floor_blocks.hip (the product's own functions compiled in isolation, one and two executions each: the
difference has no prologue, epilogue or loop control in it).  python3 tools/valu_floor.py > profiles/r05_valu_floor.json
bench.py prices the live work counters with these numbers (roofline.valu_floor_per_path)."""
import json
import os
import re
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-enable-post-misched=0",
         f"-I{REPO}/include", f"-I{REPO}/corona-13_amd/host", f"-I{REPO}/corona-13_amd/csrc", "--cuda-device-only", "-S", "-x", "hip"]


def main():
    asm = subprocess.check_output([CLANG, *FLAGS, str(REPO / "tools/micro/floor_blocks.hip"), "-o", "-"], text=True)
    kernels = {}
    for chunk in re.split(r"\n(?=_Z\w+:)", asm):
        name = chunk.split(":")[0]
        if not name.startswith("_Z") or "fb_" not in name:
            continue
        pretty = subprocess.check_output(["c++filt", name], text=True).strip()
        m = re.match(r"void (fb_\w+)<([^>]*)>", pretty)
        if not m:
            continue
        body = chunk.split(".Lfunc_end")[0]
        ins = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        kernels[(m.group(1), tuple(x.strip() for x in m.group(2).split(",")))] = {
            "valu": sum(1 for i in ins if i.startswith("v_")), "salu": sum(1 for i in ins if i.startswith("s_") and not i.startswith(("s_waitcnt", "s_nop"))),
            "lds": sum(1 for i in ins if i.startswith("ds_")), "vmem": sum(1 for i in ins if i.startswith(("global_", "flat_", "buffer_", "scratch_")))}

    def per_exec(kernel, *rest):
        a, b = kernels[(kernel, ("1",) + rest)], kernels[(kernel, ("2",) + rest)]
        return {k: b[k] - a[k] for k in a}

    blocks = {"node_visit": per_exec("fb_node_visit"), "prim_test": per_exec("fb_prim_test"), "generate": per_exec("fb_generate"),
              "vertex_all_branches": per_exec("fb_vertex", "false"), "vertex_ptdl_all_branches": per_exec("fb_vertex", "true"),
              "eval_diffuse": per_exec("fb_eval", "0"), "eval_dielectric": per_exec("fb_eval", "1"), "eval_metal": per_exec("fb_eval", "2"),
              "shadow_resolve": per_exec("fb_shadow"), "splat_pass_of_four": per_exec("fb_splat"), "sample_diffuse": per_exec("fb_sample", "0"), "sample_dielectric": per_exec("fb_sample", "1"),
              "sample_metal": per_exec("fb_sample", "2"),
              "setup_all_kinds": per_exec("fb_setup", "0"), "setup_sphere": per_exec("fb_setup", "1"), "setup_line": per_exec("fb_setup", "2"), "setup_quad": per_exec("fb_setup", "4")}
    v = blocks["vertex_all_branches"]["valu"]
    d, t, me = blocks["sample_diffuse"]["valu"], blocks["sample_dielectric"]["valu"], blocks["sample_metal"]["valu"]
    sa = blocks["setup_all_kinds"]["valu"]
    out = {"what": "vector instructions one execution of a block needs (static count, all lanes at work): two executions minus one of tools/micro/floor_blocks.hip",
           "blocks": blocks,
           # a vertex runs ONE bsdf's sample block and the surface set-up of ONE kind of primitive: the all-branches count minus what it does not run
           "vertex_valu": {"diffuse_on_quad": v - t - me - sa + blocks["setup_quad"]["valu"], "dielectric_on_line": v - d - me - sa + blocks["setup_line"]["valu"],
                           "dielectric_on_sphere": v - d - me - sa + blocks["setup_sphere"]["valu"], "metal_on_line": v - d - t - sa + blocks["setup_line"]["valu"]},
           # ptdl: the same vertex with next event estimation (nee_sample + the bsdf and pdf of the connection; both bsdfs' brdf / pdf code is counted)
           "vertex_valu_ptdl": {"diffuse_on_quad": blocks["vertex_ptdl_all_branches"]["valu"] - t - me - sa + blocks["setup_quad"]["valu"] - blocks["eval_dielectric"]["valu"] - blocks["eval_metal"]["valu"],
                                "dielectric_on_line": blocks["vertex_ptdl_all_branches"]["valu"] - d - me - sa + blocks["setup_line"]["valu"] - blocks["eval_diffuse"]["valu"] - blocks["eval_metal"]["valu"]},
           "note": "vertex = path_shade of the pt sampler between two rays (surface set-up, material ops, emitter hit, Russian roulette, bsdf sample, next ray) "
                   "with the bsdf blocks and the set-up of the primitive kinds it does not run taken out; the emitter / roulette / nested-media branches a given "
                   "vertex skips are still counted, so the vertex figures are upper estimates and the floor errs on the high side"}
    # the scene's vertex mix (classes of the surface vertices of regression/0010_pt, tools/block_probe.py of the build without the exchange) is a property of
    # the paths, not of the kernels: carried over from the newest committed floor file that has it (bench.py needs it for the floor of the bench line)
    import glob
    for prev in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", "r*_valu_floor.json")), reverse=True):
        mix = json.load(open(prev)).get("mix")
        if mix:
            out["mix"] = mix
            break
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
