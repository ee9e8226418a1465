#!/usr/bin/env python3
"""Long seeded fuzz of the device code on the mock runtime (CPU only; tests/hostsim): the suite's own fuzz drivers -- kernel-level DP
cases vs the reference's ksw_* (tests/test_dp_fuzz.py) and random mem_opt_t draws through the whole hot path vs the reference's
mem_align1_core (tests/test_opt_fuzz.py) -- with fresh seeds, round after round, under the default configuration and under the
switchable kernel forms (library options, set per handle and batch).  A failure prints the seed and the
configuration and the campaign goes on; the summary line at the end counts them.

usage: mock_fuzz_campaign.py [--minutes M] [--seed0 S] [--only dp|opt]
"""
import argparse
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DP_SETS = [{}, {"dedup_blk": 0}, {"seedsw_lds": 0}, {"ext_blk": 0}]
OPT_SETS = [
    {},
    {"chain_regs": 0},
    {"chain_regs": 1},
    {"seed_mrg": 0},
    {"seed_mrg": 2, "seed_lds_ent": 2},
    {"seed_budget": 150, "seed_p2_cap": 2},
    {"seed_tasks": 0, "publish_blk": 0, "seedsw_lds": 0, "dedup_blk": 0},
    {"seed_task_stack": 2, "seed_lds_ent": 3},
    {"occ32": 0},
    {"ptab_m": 6, "seed_lds_ent": 3},
    {"ext_occ": 4},
    {"ext_pack": 1},
    {"ext_blk": 0},
    {"dedup_heavy": 0},
    {"dedup_heavy": 2, "dedup_stage": 8, "dedup_big": 24},
    {"dedup_heavy": 2, "dedup_stage": 64, "dedup_net": 4},
    {"dedup_heavy": 2, "dedup_stage": 0},
]
LAYOUT = ("occ32", "occ32_sb_shift", "ptab_m")     # applied when the index is laid out: such a set gets a handle of its own


def with_options(dev, sets, fn):
    """Run fn with the per-batch options of `sets` set on the handle (bwagpu_set_option), then put the old values back."""
    old = {k: dev.get_option(k) for k in sets if k not in LAYOUT}
    for k in old:
        dev.set_option(k, sets[k])
    try:
        return fn()
    finally:
        for k, v in old.items():
            dev.set_option(k, v)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=30)
    ap.add_argument("--seed0", type=int, default=100000)
    ap.add_argument("--only", choices=("dp", "opt"))
    args = ap.parse_args()
    import hostsim_build
    import refapi
    import test_dp_fuzz as dp
    import test_opt_fuzz as of
    import testdata
    from bwa_amd.api import BwaGpu
    lib = hostsim_build.build()
    prefix, g = testdata.small_index()
    ref = refapi.RefIndex(prefix)
    sim = BwaGpu(prefix, lib_path=lib)
    t_end = time.time() + args.minutes * 60
    fails, rounds, seed = [], 0, args.seed0

    def attempt(what, env, fn):
        try:
            with_options(sim, env, fn)
        except Exception as e:
            if isinstance(e, AssertionError) and (not str(e) or str(e).startswith("no case")):     # the drivers' own coverage checks ("no case took the
                print(f"note {what} seed {seed}: a draw without one of the covered situations", flush=True)      # diagonal shortcut"): not a difference
                return
            fails.append((what, env, seed))
            print(f"FAIL {what} seed {seed} env {env}: {type(e).__name__}: {str(e)[:600]}", flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc()

    while time.time() < t_end:
        seed += 1
        rounds += 1
        if args.only != "opt":
            env = DP_SETS[rounds % len(DP_SETS)]
            attempt("extend", env, lambda: dp.run_extend(sim, 0, 400, 200, seed, need_stale=False))
            attempt("extend_ring", env, lambda: dp.run_extend(sim, 1, 150, 400, seed, need_stale=False, very_wide=6))
            attempt("global_lds", env, lambda: dp.run_global(sim, 2, 200, 160, 192, seed))
            attempt("global_ring", env, lambda: dp.run_global(sim, 3, 100, 200, 1 << 30, seed))
            attempt("global_ring_wide", env, lambda: dp.run_global(sim, 3, 32, 900, 1 << 30, seed + 1, wide=(63, 64, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385)))
            attempt("global_long", env, lambda: dp.run_global(sim, 5, 40, 500, 1900, seed))
            attempt("align2", env, lambda: dp.run_align2(sim, 40, seed))
        if args.only != "dp":
            env = OPT_SETS[rounds % len(OPT_SETS)]

            def opt_round():
                own = any(k in LAYOUT for k in env)
                dev = BwaGpu(prefix, lib_path=lib, options={k: v for k, v in env.items() if k in LAYOUT}) if own else sim
                try:
                    of.run_region_fuzz(dev, ref, g, draws=12, n_short=24, n_long=2, long_len=1500, seed=seed)
                finally:
                    if own:
                        dev.close()
            attempt("regions", env, opt_round)
        print(f"round {rounds} (seed {seed}) done, {len(fails)} failure(s), {(t_end - time.time()) / 60:.1f} min left", flush=True)
    print(f"SUMMARY: {rounds} rounds, {len(fails)} failures: {fails}", flush=True)
    sim.close(); ref.close()
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
