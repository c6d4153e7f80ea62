#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loops, for pricing its VALU instructions by width (bench.py, roofline.valu).

The hardware counts double-precision adds, multiplies, FMAs and transcendentals (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64) but no
counter separates v_min_f64 / v_max_f64 / v_cmp_*_f64 (scripts/ubench/valu_class.hip run under those counters shows them in none of
the per-type counters).  They run at the fp64 rate too, so bench.py estimates their number as ADD_F64 x (their static count per
v_add_f64 in the basic blocks of the kernel that are BP passes: blocks with at least three LDS instructions and three fp64 instructions).

    python scripts/isa_mix.py <tag> <workload>      compiles the workload's translation unit with --save-temps into /tmp/isa_mix and
                                                    writes profiles/<tag>_<workload>_isa_mix.json"""
import json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "slidingwindowdecoder_amd", "csrc")
# workload -> (translation unit, mangled-name fragment of its kernel)
KERNELS = {
    "headline": ("swd_kernels_k3.hip", "pipeline_kernelILi256ELi7ELi6ELi9ELi3ELb0ELb0ELi7E"),
    "bb288": ("swd_kernels_k3.hip", "pipeline_kernelILi1024ELi5ELi6ELi6ELi3ELb1ELb0ELi5E"),
    "gdg": ("swd_kernels_k2.hip", "pipeline_kernelILi256ELi7ELi6ELi9ELi2ELb0ELb0ELi2E"),
    "gdg64": ("swd_kernels_k7.hip", "pipeline_kernelILi256ELi7ELi6ELi9ELi7ELb0ELb0ELi2E"),
    "global144": ("swd_kernels_k5.hip", "pipeline_kernelILi1024ELi9ELi6ELi9ELi3ELb0ELb1ELi9E"),
    "bp4": ("swd_bp4.hip", "bp4_kernelILi4ELi4E"),
}
FLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-sched-strategy=iterative-ilp "
         "-mllvm -amdgpu-atomic-optimizer-strategy=None -I../../include -I. --cuda-device-only -S").split()


def scan(asm_path, frag, min_ds=3):
    blocks, cur, infn = [], None, False
    for ln in open(asm_path):
        ln = ln.rstrip("\n")
        if not infn:
            if ln.startswith("_ZN3swd") and frag in ln and ln.split(";")[0].rstrip().endswith(":"):
                infn, cur = True, ["entry", []]
                blocks.append(cur)
            continue
        if "s_endpgm" in ln:
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = [m.group(1), []]
            blocks.append(cur)
        elif ln.startswith("\t") and not ln.strip().startswith((";", ".")):
            cur[1].append(ln.strip())
    tot = {"v_add_f64": 0, "v_mul_f64": 0, "v_fma_f64": 0, "f64_minmaxcmp": 0, "valu": 0, "ds": 0, "scratch": 0, "blocks": 0}
    for _, b in blocks:
        nds = sum(x.startswith("ds_") for x in b)
        f64 = [x for x in b if re.match(r"v_(add|mul|fma|min|max|cmp\w*|cmpx\w*)_\w*f64", x)]
        if nds < min_ds or len(f64) < 3:
            continue
        tot["blocks"] += 1
        tot["ds"] += nds
        tot["scratch"] += sum(x.startswith("scratch_") for x in b)
        tot["valu"] += sum(x.startswith("v_") for x in b)
        for x in f64:
            op = x.split()[0]
            if op.startswith("v_add_f64"): tot["v_add_f64"] += 1
            elif op.startswith("v_mul_f64"): tot["v_mul_f64"] += 1
            elif op.startswith("v_fma_f64"): tot["v_fma_f64"] += 1
            else: tot["f64_minmaxcmp"] += 1
    return tot, len(blocks)


def main():
    tag, wl = sys.argv[1], sys.argv[2]
    tu, frag = KERNELS[wl]
    out_dir = "/tmp/isa_mix"
    os.makedirs(out_dir, exist_ok=True)
    asm = os.path.join(out_dir, tu.replace(".hip", ".s"))
    if not os.path.exists(asm) or os.path.getmtime(asm) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [tu, "-o", asm], cwd=CSRC)
    tot, nblocks = scan(asm, frag, 1 if wl == "bp4" else 3)  # (bp4: the per-edge loops of its passes are rolled -- one LDS access per block)
    if not tot["blocks"]:
        raise SystemExit(f"no BP-pass blocks found for {frag} in {asm}")
    res = {"workload": wl, "kernel_fragment": frag, "translation_unit": tu, "basic_blocks": nblocks, "bp_pass_blocks": tot["blocks"],
           "static_counts_in_bp_pass_blocks": tot,
           "f64_minmaxcmp_per_f64_add": tot["f64_minmaxcmp"] / max(tot["v_add_f64"], 1),
           "note": "static counts over the basic blocks that are BP passes (>= 3 LDS and >= 3 fp64 instructions); bench.py multiplies the "
                   "hardware's SQ_INSTS_VALU_ADD_F64 by this ratio to estimate the fp64 min / max / compare instructions no counter separates"}
    json.dump(res, open(os.path.join(ROOT, "profiles", f"{tag}_{wl}_isa_mix.json"), "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
