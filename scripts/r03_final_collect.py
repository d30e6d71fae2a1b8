"""Copies the summaries of scripts/r03_final.sh from gpurun_out/r03f into profiles/ (with a provenance header) and derives
profiles/r03_pmc_traffic.json (HBM bytes per launch of the attention kernels = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE doubled
per the gfx950 correction of MI355X_MICROARCH.md) that bench.py reports as roofline.traffic."""
import json, os, re, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r03f"), os.path.join(ROOT, "profiles")
HEAD = {
    "r03_bench_direct128_rocprofv3_kernel_stats.txt": "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-extra --no-cpu-baseline (round-3 closing build; scripts/r03_final.sh step 1); the bench line of that run: profiles/r03_bench_direct128_under_rocprof.json\n",
    "r03_pmc_fetch_write_bench_direct128.txt": "# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no trace domains) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra; per-kernel averages in KiB (scripts/pmc_summary.py); FETCH_SIZE is NOT yet doubled here\n",
    "r03_pmc_attention_selfattn_N32768_p0.1.txt": "# rocprofv3 --pmc (two passes, 8 SQ counters each) -- python3 scripts/attn_only.py 128 0.1: self-attention B2 H4 N32768 D64 bf16, dropout 0.1, 3 launches each; round-3 closing kernels (8-wavefront dQ / dK/dV workgroups, LDS-DMA tile loads; 'attn_fw' = attn_fwd2_kernel<64, true, 4>)\n",
    "r03_gemm_vs_hipblaslt.txt": "# scripts/gemm_vs_blas.py, round-3 closing build (persistent tile walk + straight-line epilogue), plain block shapes, hvc_gemm vs torch.mm (hipBLASLt: MT256x256x32 persistent / stream-K kernels, see DESIGN.md)\n",
    "r03_gemm_shapes_direct128.txt": "# scripts/gemm_shapes.py direct128, round-3 closing build: every ops.gemm shape of one 128^3 train step (round 2: profiles/r02_gemm_shapes_direct128.txt, 5.44 ms/step)\n",
    "r03_attention_fp8_x16_vs_bf16.txt": "# scripts/attn_fp8_bench.py, round-3 closing build, default fp8 kernel (x16); the x64 experiment kernel: profiles/r03_attention_fp8_x64_experiment.txt\n",
    "r03_cascade_stage3_256_rocprofv3_kernel_stats.txt": "# rocprofv3 --kernel-trace --stats -- python3 scripts/cascade_fullsize.py 3 1 4 (cascade stage 3 at 256^3, B = 1, checkpoint policy auto = off; steps 182.5 - 185 ms, see r03_cascade_stage3_256_steps.log)\n",
}
for name, head in HEAD.items():
    body = open(os.path.join(SRC, name)).read()
    body = "\n".join(l for l in body.split("\n") if "amdgpu.ids" not in l)
    open(os.path.join(DST, name), "w").write(head + body)
shutil.copy(os.path.join(SRC, "bench_direct128_under_rocprof.json"), os.path.join(DST, "r03_bench_direct128_under_rocprof.json"))
shutil.copy(os.path.join(SRC, "cascade3_steps.log"), os.path.join(DST, "r03_cascade_stage3_256_steps.log"))
shutil.copy(os.path.join(SRC, "bench_default.json"), os.path.join(DST, "r03_bench_default.json"))
# traffic json
cur, vals = None, {}
for l in open(os.path.join(SRC, "r03_pmc_fetch_write_bench_direct128.txt")):
    if l.startswith("=="):
        cur = l[2:].strip()
    else:
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", l)
        if m and cur:
            vals.setdefault(cur, {})[m.group(1)] = float(m.group(3))
def traffic(needle):
    k = next(k for k in vals if needle in k)
    return (2 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024
out = {"_note": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in two separate passes of "
                "`bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra` (round-3 closing kernels; profiles/r03_pmc_fetch_write_bench_direct128.txt), "
                "averaged over all launches of the kernel (self- and cross-attention shapes), FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section)",
       "direct128": {"attn_fwd2_kernel": traffic("attn_fw"), "attn_fwd_kernel": traffic("attn_fw"),
                     "attn_bwd_dkv_kernel": traffic("attn_bwd_dkv_kernel"), "attn_bwd_dq_kernel": traffic("attn_bwd_dq_kernel")}}
json.dump(out, open(os.path.join(DST, "r03_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["direct128"]))
