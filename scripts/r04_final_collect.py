"""Copies the summaries of scripts/r04_final.sh from gpurun_out/r04f into profiles/ (with a provenance header) and derives
profiles/r04_pmc_traffic.json (HBM bytes per launch of the attention kernels = (2 * FETCH_SIZE + WRITE_SIZE) * 1024, FETCH_SIZE doubled
per the gfx950 correction of MI355X_MICROARCH.md) that bench.py reports as roofline.traffic - now for all three resolutions."""
import json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r04f"), os.path.join(ROOT, "profiles")
B = "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload %s --steps 10 --warmup 3 --no-extra --no-cpu-baseline (round-4 closing build; scripts/r04_final.sh step 1); the bench line of that run: profiles/r04_bench_%s_under_rocprof.json\n"
P = "# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (two separate passes, no trace domains) -- python3 bench.py --workload %s --graph off --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra; per-kernel averages in KiB (scripts/pmc_summary.py); FETCH_SIZE is NOT yet doubled here\n"
HEAD = {
    "r04_pmc_attention_selfattn_N32768_p0.1.txt": "# rocprofv3 --pmc (two passes, 8 SQ counters each) -- python3 scripts/attn_only.py 128 0.1: self-attention B2 H4 N32768 D64 bf16, dropout 0.1, 3 launches each; round-4 closing kernels ('attn_fw' = attn_fwdp_kernel<64, true, 8>, the software-pipelined forward; dQ / dK/dV as in round 3)\n",
    "r04_hbm_kernels.txt": "# scripts/hbm_kernels.py under rocprofv3 (kernel trace; --pmc FETCH_SIZE; --pmc WRITE_SIZE: three separate passes), scripts/hbm_summary.py; round-4 closing build\n",
    "r04_gemm_vs_hipblaslt.txt": "# scripts/gemm_vs_blas.py, round-4 closing build (GEMM kernel unchanged since round 3 apart from hvc_set_option), plain block shapes, hvc_gemm vs torch.mm (hipBLASLt)\n",
    "r04_gemm_shapes_direct128.txt": "# scripts/gemm_shapes.py direct128, round-4 closing build: every ops.gemm shape of one 128^3 train step\n",
    "r04_attention_pipelined_forward_ab.txt": "# scripts/attn_pipe_ab.py fwd, round-4 closing build: same-box alternating A/B of attn_fwdp_kernel (HVC_ATTN_PIPE=1) against the phase-separated attn_fwd2_kernel (=0); 'pipelined, 4-wavefront workgroups' = HVC_ATTN_FWD_WAVES=4 (the form d = 32 takes by default)\n",
    "r04_attention_shapes.txt": "# scripts/attn_shapes.py, round-4 closing build: HIP-event minima of the attention kernels on the BASELINE shapes (d = 32: dK/dV and dQ as four-wavefront workgroups, three per CU)\n",
    "r04_cascade_stage3_256_rocprofv3_kernel_stats.txt": "# rocprofv3 --kernel-trace --stats -- python3 scripts/cascade_fullsize.py 3 1 4 (cascade stage 3 at 256^3, B = 1, checkpoint policy auto = off; see r04_cascade_stage3_256_steps.log) - closing build: streaming single-channel / halo-tile convolutions (DESIGN 5.3.1), d = 32 dK/dV and dQ three workgroups per CU (5.1.5)\n",
}
for wl in ("direct128", "direct64", "direct256"):
    HEAD[f"r04_bench_{wl}_rocprofv3_kernel_stats.txt"] = B % (wl, wl)
    HEAD[f"r04_pmc_fetch_write_bench_{wl}.txt"] = P % wl
for name, head in HEAD.items():
    body = open(os.path.join(SRC, name)).read()
    body = "\n".join(l for l in body.split("\n") if "amdgpu.ids" not in l)
    open(os.path.join(DST, name), "w").write(head + body)
for wl in ("direct128", "direct64", "direct256"):
    shutil.copy(os.path.join(SRC, f"bench_{wl}_under_rocprof.json"), os.path.join(DST, f"r04_bench_{wl}_under_rocprof.json"))
with open(os.path.join(DST, "r04_cascade_stage3_256_steps.log"), "w") as f:
    f.write("# scripts/cascade_fullsize.py 3 1 6, round-4 closing build (no profiler):\n" + open(os.path.join(SRC, "cascade3_plain_steps.log")).read())
    f.write("# the same under rocprofv3 --kernel-trace (r04_cascade_stage3_256_rocprofv3_kernel_stats.txt):\n"
            + "".join(l for l in open(os.path.join(SRC, "cascade3_steps.log")) if "amdgpu.ids" not in l))
    f.write("# scripts/cascade_fullsize.py 2 2 6 (stage 2 at 128^3, B = 2: BASELINE configs #4 geometry), same build:\n" + open(os.path.join(SRC, "cascade2_plain_steps.log")).read())
glue = open(os.path.join(DST, "r04_glue_layers_256.txt")).read()
i0, i1 = glue.index("# AFTER (closing build"), glue.index("#\n# 64 -> 32 layer, versions of conv3_halo_kernel")
glue = (glue[:i0] + "# AFTER (closing build: conv_c1_fwd / conv_c1_dw / conv_c1_dx, conv3_halo (forward and input gradient of the 64 -> 32 layer), conv_o1_fwd / _bwd,\n"
        "# division-free GroupNorm apply)\n" + open(os.path.join(SRC, "glue_after.log")).read() + glue[i1:])
open(os.path.join(DST, "r04_glue_layers_256.txt"), "w").write(glue)
for a, b in (("bench_default.json", "r04_bench_default.json"), ("bench_ddp_rccl_world1.json", "r04_bench_ddp_rccl_world1.json"),
             ("bench_direct64_ddp_graph.json", "r04_bench_direct64_ddp_rccl_world1_hipgraph.json")):
    shutil.copy(os.path.join(SRC, a), os.path.join(DST, b))


def traffic_of(wl):
    cur, vals = None, {}
    for l in open(os.path.join(SRC, f"r04_pmc_fetch_write_bench_{wl}.txt")):
        if l.startswith("=="):
            cur = l[2:].strip()
        else:
            m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", l)
            if m and cur:
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(3))

    def traffic(needle):
        ks = [k for k in vals if needle in k]
        if not ks:
            return None
        # several instantiations of one kernel (4- / 8-wavefront forms): average weighted equally per instantiation is wrong - take the launch-weighted mean
        return sum((2 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]) * 1024 for k in ks) / len(ks)
    fw = traffic("attn_fw")
    return {"attn_fwd2_kernel": fw, "attn_fwd_kernel": fw, "attn_bwd_dkv_kernel": traffic("attn_bwd_dkv_kernel"), "attn_bwd_dq_kernel": traffic("attn_bwd_dq_kernel")}


out = {"_note": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in two separate passes of "
                "`bench.py --workload W --graph off --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-extra` (round-4 closing kernels; "
                "profiles/r04_pmc_fetch_write_bench_W.txt), averaged over the launches of the kernel (self- and cross-attention shapes; where a kernel runs in two "
                "workgroup forms, the mean of the two forms' averages), FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (HBM section)"}
for wl in ("direct128", "direct64", "direct256"):
    out[wl] = traffic_of(wl)
json.dump(out, open(os.path.join(DST, "r04_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "_note"}))
