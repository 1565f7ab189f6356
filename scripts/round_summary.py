"""profiles/r05_summary.md from the evidence files scripts/profile_round.sh left (copied to profiles/ as r05_*; the r04 page was written by the r4 version of this
script: git history).  usage: python scripts/round_summary.py"""
import csv
import json
import os

R = "r05"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles") + "/"


def b(n):
    return json.load(open(P + "%s_bench_%s.json" % (R, n)))


def ks(n, sub):
    for r in csv.DictReader(open(P + "%s_kernel_stats_%s.csv" % (R, n))):
        if sub in r["Name"]:
            return float(r["AverageNs"]) / 1e6, int(r["Calls"])


c2, c2s, c4, c4s, c1, fr, fg, sx, sx4 = (b(n) for n in ("cfg2", "cfg2_sympy", "cfg4", "cfg4_sympy", "cfg1", "fv-ref", "fv-grid", "cfg2_self_exchange", "cfg4_self_exchange"))
a2, b2 = ks("cfg2", "dg_stage_a_reg_kernel"), ks("cfg2", "dg_stage_b")
a2s, b2s = ks("cfg2_sympy", "dg_stage_a_reg_kernel"), ks("cfg2_sympy", "dg_stage_b")
a4, k1, kf = ks("cfg4", "dg_stage_a_m8"), ks("cfg1", "dg_fused_single"), ks("fv-ref", "fv_rusanov_kernel")
kg4, kg15 = ks("fv-grid", "FvShape<4, 1, 5, 10>, true, true>"), ks("fv-grid", "slab_kernel<exa::Euler, 1, true, false, true>")
t = {n: json.load(open(P + f)) for n, f in (("cfg2", "stage_a_traffic.json"), ("cfg4", "traffic_cfg4.json"), ("cfg1", "traffic_cfg1.json"), ("fvref", "traffic_fv_ref.json"),
                                             ("g4", "traffic_fv_grid_4x4.json"), ("g15", "traffic_fv_grid_15.json"))}
m8 = json.load(open(P + "m8_pmc.json"))
sh = fg["shapes"]
g4, g15 = sh["ref-4x4"], sh["limiter-15^3"]
out = f"""# Round 5 -- every benchmarked kernel against its roof (one MI355X, final build)

Source of every number: `scripts/profile_round.sh r05` (bench lines, `rocprofv3 --kernel-trace --stats`, FETCH_SIZE / WRITE_SIZE in separate `--pmc` passes,
FETCH x2 per the gfx950 correction, SQ counters in passes of their own) -- files `r05_bench_*.json`, `r05_kernel_stats_*.csv`, `traffic_*.json`, `stage_a_traffic.json`,
`stage_a_pmc.json`, `m8_pmc.json` beside this one (this page: `scripts/round_summary.py`); notes: `r05_m8_kernel.txt` (p = 7 stage A: what moved it, what did not),
`r05_valu_lds_overlap.txt` (do fp64 vector, matrix and LDS instructions overlap on a CU?), `r05_xt_ncp_kernels.txt`.  Roofs: fp64 78.6 TFLOP/s, HBM 8.0 TB/s.

| configuration | step | dominant kernel | launch (rocprofv3 average) | roof | frac | HBM traffic vs algorithmic | round 4 |
|---|---|---|---|---|---|---|---|
| cfg 2: 3-D Euler p = 5, 128^3 cells (headline) | {c2['ms_per_step']:.1f} ms, {c2['value']:.3e} DoF-upd/s | `dg_stage_a_reg_kernel<6, Euler, 2, false>` | {a2[0]:.2f} ms ({a2[1]} calls); events in bench {c2['roofline']['launch_ms']:.2f} | fp64 | **{c2['roofline']['frac']:.3f}** | {t['cfg2']['hbm_bytes_per_launch']/1e9:.3f} GB = {t['cfg2']['hbm_bytes_per_launch']/t['cfg2']['algorithmic_bytes_per_launch']:.5f}x | 170.6 ms, 0.422 (kernel closed: DESIGN 4.1e, r5 table) |
| ... stage B | | `dg_stage_b_dense_kernel<3, 6, Euler, 1, 256>` | {b2[0]:.2f} ms | HBM stream | at the stream rate | | 15.5 ms |
| **cfg 2 on the SymPy-specified Euler** (`other_configs.cfg2_sympy`) | {c2s['ms_per_step']:.1f} ms, {c2s['value']:.3e} = **{c2s['value']/c2['value']:.3f}x** the built-in | `dg_stage_a_reg_kernel<6, UserPDE, 2, false>` | {a2s[0]:.2f} ms; stage B **{b2s[0]:.2f} ms** | fp64 | **{c2s['roofline']['frac']:.3f}** | | 0.964x; stage B 17.48 ms (`switch (d)` in the generated eigenvalue) |
| cfg 4: 3-D Euler p = 7 + limiter, 64^3 cells | **{c4['ms_per_step']:.1f} ms, {c4['value']:.3e}** | `dg_stage_a_m8_kernel<Euler>` | **{a4[0]:.2f} ms** | fp64 (vector + matrix instructions: one pipe) | **{c4['roofline']['frac']:.3f}** (matrix instructions busy {m8['mfma_busy']:.3f} of the SIMD time, 16 cycles each) | {t['cfg4']['hbm_bytes_per_launch']/1e9:.1f} GB = {t['cfg4']['hbm_bytes_per_launch']/18.79e9:.2f}x | 107.5 ms, 6.25e9; stage A 97.0 ms = 0.435 |
| **cfg 4 on the SymPy-specified Euler** | {c4s['ms_per_step']:.1f} ms, {c4s['value']:.3e} = **{c4s['value']/c4['value']:.3f}x** | `dg_stage_a_m8_kernel<UserPDE>` | | fp64 | **{c4s['roofline']['frac']:.3f}** | | 0.973x |
| cfg 1: 2-D Euler p = 3, 512^2, single stage | {c1['ms_per_step']:.3f} ms, {c1['value']:.3e} | `dg_fused_single_kernel<4, Euler, 4, 4>` | {k1[0]:.3f} ms | HBM (algorithmic bytes) | **{c1['roofline']['frac']:.3f}** (of its own compulsory traffic: {c1['roofline']['frac_of_compulsory']:.3f}) | {t['cfg1']['hbm_bytes_per_launch']/1e9:.2f} GB (fused) | 0.138 ms |
| fv-ref: reference configuration, 2^20 patches, faithful, in place | {fr['ms_per_step']:.3f} ms, {fr['value']:.3e} | `fv_rusanov_kernel<2, EulerRef2D, 0, ..., FvShape<4,1,5,10>, true, false>` | {kf[0]:.3f} ms | HBM | **{fr['roofline']['frac']:.3f}** | {t['fvref']['hbm_bytes_per_launch']/1e9:.2f} GB = {t['fvref']['hbm_bytes_per_launch']/t['fvref']['algorithmic_bytes_per_launch']:.2f}x | 0.460 |
| fv-grid: 2^20 patches 4x4 as a periodic grid, halo + update + CFL per step (halo-less arrays) | {g4['step_with_cfl_ms']:.3f} ms with the CFL read = **{g4['step_with_cfl_over_bare']:.2f}x** the bare kernel ({g4['bare_kernel_ms']:.3f} ms) | `fv_rusanov_kernel<..., FvShape<4,1,5,10>, true, GRID>` | {kg4[0]:.3f} ms | HBM | {g4['frac_of_hbm_peak']:.3f} of 16 V B per volume | {t['g4']['hbm_bytes_per_launch']/1e9:.2f} GB = {t['g4']['hbm_bytes_per_launch']/t['g4']['algorithmic_bytes_per_launch']:.2f}x | 0.87x, 1.07x |
| **fv-grid: 8 192 patches 15^3** | {g15['step_with_cfl_ms']:.3f} ms = **{g15['step_with_cfl_over_bare']:.2f}x** ({g15['bare_kernel_ms']:.3f} ms) | `fv_rusanov_slab_kernel<Euler, 1, true, false, GRID>` | {kg15[0]:.3f} ms | HBM | {g15['frac_of_hbm_peak']:.3f} | **{t['g15']['hbm_bytes_per_launch']/1e9:.2f} GB = {t['g15']['hbm_bytes_per_launch']/t['g15']['algorithmic_bytes_per_launch']:.2f}x** (XCD-contiguous patch order) | 1.09x; 1.32x the algorithmic bytes |

Exchange rehearsals (RCCL send / recv to self, one GPU): cfg 2 sharded step (`r05_bench_cfg2_self_exchange.json`): RCCL span {sx['exchange_ms']:.2f} ms, pack {sx['pack_ms']:.2f} ms, exposed
{sx['exposed_exchange_ms']:.2f} ms, step {sx['ms_per_step']:.1f} ms, per-rank roofline.frac {sx['roofline']['frac']:.3f} over {sx['roofline']['launches_per_step']} stage-A launches per step; **cfg 4 sharded
limited step** (new: `bench.py --config cfg4 --gpus N`, `r05_bench_cfg4_self_exchange.json`): step {sx4['ms_per_step']:.1f} ms, trace exchange {sx4['exchange_ms']:.2f} ms (exposed {sx4['exposed_exchange_ms']:.2f}),
the limiter's flag + subcell-layer exchanges {sx4['limiter_exchange_ms']:.2f} ms, per-rank roofline.frac {sx4['roofline']['frac']:.3f}.  `reserve_cus` trial in the warm-up (cfg 2): {json.dumps(sx.get('reserve_cus_trial'))} -> {sx.get('reserve_cus_chosen')} kept.
No multi-GPU run exists (no node was available in any round): these lines exercise the code path and its measurement fields, they are not a scaling curve.
"""
open(P + "%s_summary.md" % R, "w").write(out)
print(out)
