"""CPU: bench.py's `variants` object end to end -- bench.run_variants starts tools/variant_probe.py as a child process per leg and collects its
JSON lines -- against the mock runtime (one stream: the mock is single-threaded), so that the code the driver's GPU run depends on has run
somewhere before it runs there.  The numbers mean nothing here; the structure and the result digests do."""
import argparse
import os
import sys

import numpy as np

import hostsim_build
import testdata

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_bench_variants_object(monkeypatch, tmp_path):
    import bench
    prefix, g = testdata.small_index()
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):          # (the probe reads <prefix>.codes.npy next to the index files)
        os.symlink(prefix + ext, str(tmp_path / ("idx" + ext)))
    np.save(str(tmp_path / "idx.codes.npy"), g)
    monkeypatch.setenv("BWA_AMD_PROBE_LIB", hostsim_build.build())
    args = argparse.Namespace(variants="seed_mrg=2;BWAGPU_SEED_LDS_ENT=3", variants_timeout=600.0, reads=8, read_len=150, streams=1, dense_sa=0,
                              no_longread=False, long_reads=1, long_len=1150)
    from bwa_amd import simdata
    short = simdata.make_reads_se(g, 8, seed=3)                   # bench.py hands its own batches over as files
    np.save(str(tmp_path / "variant_batch0.npy"), short)
    np.save(str(tmp_path / "long_reads.npy"), simdata.make_reads_long(g, 1, length=1150, seed=4))
    res = bench.run_variants(args, str(tmp_path / "idx"), [str(tmp_path / "variant_batch0.npy")])
    short, long_ = res["short_reads"], res["long_reads"]
    assert short["rc"] == 0 and long_["rc"] == 0, (short, long_)
    assert [r["config"] for r in short["runs"]] == ["defaults", "seed_mrg=2", "BWAGPU_SEED_LDS_ENT=3"]      # (the round-3 spelling of an option is still accepted)
    assert [r["config"] for r in long_["runs"]] == ["defaults", "seed_mrg=0 seed_tasks=0 publish_blk=0 seedsw_lds=0 dedup_blk=0",
                                                    "seed_mrg=0", "seed_tasks=0", "publish_blk=0", "seedsw_lds=0", "dedup_blk=0"]
    for r in short["runs"] + long_["runs"]:
        assert "error" not in r and r["same_result_as_defaults"] is True, r
    assert all("ms_per_step" in r and "stage_ms_solo" in r for r in short["runs"]) and all("ms_per_pass" in r for r in long_["runs"])


def test_run_child_keeps_what_a_hung_child_printed():
    """A variant that hangs must cost the bench line nothing but its own entries: the child is killed with its process group at the time
    limit, the lines it had printed are kept, and the call returns at once."""
    import sys
    import time
    import bench
    leg = {}
    t = time.time()
    text, rc = bench.run_child([sys.executable, "-c", "import time,sys; print('{\"config\": \"x\"}', flush=True); sys.stderr.write('boom'); sys.stderr.flush(); time.sleep(60)"], 1.5, leg)
    assert rc == "timeout" and '{"config": "x"}' in text and time.time() - t < 10 and "boom" in leg.get("stderr_tail", "") and not leg.get("unreaped")
    text, rc = bench.run_child([sys.executable, "-c", "print('ok')"], 10, leg)
    assert rc == 0 and text.strip() == "ok"


def test_variants_share_the_wall_budget(tmp_path):
    """The bench line must come out within --wall-budget: with little of it left no child is started at all, and the first leg never
    gets more than 55 % of what is left."""
    import time
    import bench
    args = argparse.Namespace(variants="seed_mrg=2", variants_timeout=600.0, reads=8, read_len=150, streams=1, dense_sa=0,
                              no_longread=False, long_reads=1, long_len=1150)
    t = time.time()
    res = bench.run_variants(args, str(tmp_path / "none"), [], wall_left=15.0)
    assert "skipped" in res["short_reads"] and "skipped" in res["long_reads"] and time.time() - t < 2
    t = time.time()
    res = bench.run_variants(args, str(tmp_path / "none"), [], wall_left=38.0)      # 20.9 s for the first leg (its probe fails at once: no index), < 20 s left for the second
    assert res["short_reads"].get("rc") not in (None, 0) and time.time() - t < 25


def test_steady_state_from_the_timeline_trace():
    """bench.steady_state_from_trace: the FASTQ -> SAM run's steady rate is the spacing of the device stage's completions past the first three batches
    (the handles' first ones, which include the pipeline's fill) and before the last (short) one."""
    import bench
    lines = ["[D::timeline] batch 0 read 0.050 .. 0.130", "[D::timeline] batch 0 device 0.150 .. 0.400 (slot 0)"]
    t = 0.5
    for b in range(1, 12):
        lines.append(f"[D::timeline] batch {b} device {t - 0.2:.3f} .. {t:.3f} (slot {b % 3})")
        lines.append(f"[D::timeline] batch {b} finalize {t:.3f} .. {t + 0.05:.3f}")
        t += 0.08
    got = bench.steady_state_from_trace("\n".join(lines), 500000)
    assert got["first_batch_out_s"] == 0.4 and abs(got["ms_per_batch"] - 80.0) < 0.2 and abs(got["Mreads_s"] - 6.25) < 0.02
    assert bench.steady_state_from_trace("\n".join(lines[:8]), 500000) is None       # too few batches
    assert bench.steady_state_from_trace("\n".join(lines), None) is None


def test_bench_run_product_parses_the_command_lines_trace(monkeypatch, tmp_path):
    """bench.run_product end to end against the mock-runtime build of `bwa-amd mem`: the rate line, the stages' busy and CPU times, the device stage's
    per-batch steps and the timeline trace (`steady_state`) are parsed from what the program really prints -- a change of a trace line's wording shows
    up here, not as a missing field in the GPU box's bench line."""
    import bench
    import test_cli
    from bwa_amd import simdata
    prefix, g = testdata.small_index()
    r1, r2 = simdata.make_reads_pe(g, 1000, seed=5)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    simdata.write_fastq(f1, r1); simdata.write_fastq(f2, r2)
    os.makedirs(str(tmp_path / "root" / "bwa_amd"))
    os.symlink(test_cli._sim_cli(), str(tmp_path / "root" / "bwa_amd" / "bwa-amd"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path / "root"))
    monkeypatch.setenv("BWAGPU_PTAB_M", "6"); monkeypatch.setenv("BWAGPU_CLI_SERIALIZE", "1")
    res = bench.run_product(prefix, [f1, f2], 2, None, streams=2, K=30000, timeout=900)
    assert res is not None and res["n"] == 2000 and res["n_batches"] == 10 and res["retries"] == 0 and res["handles"] == 2
    dev = res["device_stage_ms_per_batch"]
    assert dev["reads_per_batch"] == 200 and dev["hot_path"] > 0 and dev["cigar_kernels"] >= 0
    ss = res["steady_state"]
    assert ss is not None and ss["ms_per_batch"] > 0 and ss["first_batch_out_s"] > 0
    assert set(res["stage_us_per_read"]) >= {"read", "encode", "device", "finalize", "write"}
    assert res["cpu_us_per_read"]["threads"] == 2
