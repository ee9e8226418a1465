#!/usr/bin/env python3
"""Long seeded fuzz of the device code on the mock runtime (CPU only; tests/hostsim): the suite's own fuzz drivers -- kernel-level DP
cases vs the reference's ksw_* (tests/test_dp_fuzz.py) and random mem_opt_t draws through the whole hot path vs the reference's
mem_align1_core (tests/test_opt_fuzz.py) -- with fresh seeds, round after round, under the default configuration and under the
switchable kernel forms.  Every switch is read per call, so one process covers them all.  A failure prints the seed and the
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

DP_ENVS = [{}, {"BWAGPU_LONG_QLDS": "1"}, {"BWAGPU_DEDUP_BLK": "1", "BWAGPU_EXT_BLK": "1"}, {"BWAGPU_EXT_BLK": "1", "BWAGPU_LONG_QLDS": "1"}]
OPT_ENVS = [
    {},
    {"BWAGPU_SEED_MRG": "1"},
    {"BWAGPU_SEED_MRG": "2", "BWAGPU_SEED_LDS_ENT": "2"},
    {"BWAGPU_SEED_MRG": "2", "BWAGPU_PUBLISH_BLK": "1", "BWAGPU_LONG_QLDS": "1", "BWAGPU_SEEDSW_LDS": "1", "BWAGPU_SEED_CHUNK": "128", "BWAGPU_DEDUP_BLK": "1", "BWAGPU_EXT_BLK": "1"},
    {"BWAGPU_SEED_CHUNK": "256", "BWAGPU_SEEDSW_LDS": "1"},
    {"BWAGPU_OCC32": "0"},
    {"BWAGPU_PTAB_M": "6", "BWAGPU_SEED_LDS_ENT": "3"},
    {"BWAGPU_CHAIN_LDS": "0"},
]
LAYOUT = ("BWAGPU_OCC32", "BWAGPU_OCC32_SB_SHIFT", "BWAGPU_PTAB_M")     # read when the index is laid out: such a set gets a handle of its own


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


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
            with_env(env, fn)
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
            env = DP_ENVS[rounds % len(DP_ENVS)]
            attempt("extend", env, lambda: dp.run_extend(sim, 0, 400, 200, seed, need_stale=False))
            attempt("extend_ring", env, lambda: dp.run_extend(sim, 1, 150, 400, seed, need_stale=False, very_wide=6))
            attempt("global_lds", env, lambda: dp.run_global(sim, 2, 200, 160, 192, seed))
            attempt("global_ring", env, lambda: dp.run_global(sim, 3, 100, 200, 1 << 30, seed))
            attempt("global_ring_wide", env, lambda: dp.run_global(sim, 3, 32, 900, 1 << 30, seed + 1, wide=(63, 64, 127, 128, 129, 200, 255, 256, 257, 300, 383, 384, 385)))
            attempt("global_long", env, lambda: dp.run_global(sim, 5, 40, 500, 1900, seed))
            attempt("align2", env, lambda: dp.run_align2(sim, 40, seed))
        if args.only != "dp":
            env = OPT_ENVS[rounds % len(OPT_ENVS)]

            def opt_round():
                own = any(k in LAYOUT for k in env)
                dev = BwaGpu(prefix, lib_path=lib) if own else sim
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
