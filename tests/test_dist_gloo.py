"""CPU, world_size 2 over gloo: the multi-GPU start-up and sharding logic (bwa_amd/dist.py) with the mock-runtime build:
rank 0 loads the index, rank 1 receives it by broadcast; both align their contiguous shard; the concatenation must equal
the single-process result.  (On the GPU box the same code runs over RCCL; see test_gpu_parity.py.)"""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

import testdata
from bwa_amd import simdata
from bwa_amd.structs import default_opt


def _worker(rank, world, port, lib_path, prefix, out_dir):
    import torch.distributed as dist
    from bwa_amd import dist as bdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gpu = bdist.broadcast_index(prefix, device=0, src=0, lib_path=lib_path)
    g, _ = testdata.small_genome()
    r1, r2 = simdata.make_reads_pe(g, 21, seed=71)
    reads = np.empty((42, 150), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
    seqs, off = testdata.flat(reads)
    lo, hi, counts, regs = bdist.align_sharded(gpu, default_opt(), seqs, off, pair=True)
    assert lo % 2 == 0 and hi % 2 == 0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=lo, hi=hi, counts=counts, regs=regs)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_broadcast_and_shard(tmp_path):
    import hostsim_build
    from bwa_amd.api import BwaGpu
    from bwa_amd.dist import shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    lib = hostsim_build.build()
    prefix, g = testdata.small_index()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, lib, prefix, str(tmp_path)), nprocs=2, join=True)
    r1, r2 = simdata.make_reads_pe(g, 21, seed=71)
    reads = np.empty((42, 150), dtype=np.uint8); reads[0::2], reads[1::2] = r1, r2
    single = BwaGpu(prefix, lib_path=lib)
    c, r = single.align(default_opt(), *testdata.flat(reads))
    parts = [np.load(os.path.join(str(tmp_path), f"rank{k}.npz")) for k in range(2)]
    assert int(parts[0]["lo"]) == 0 and int(parts[0]["hi"]) == int(parts[1]["lo"]) and int(parts[1]["hi"]) == 42
    assert np.array_equal(np.concatenate([p["counts"] for p in parts]), c)
    assert np.concatenate([p["regs"] for p in parts]).tobytes() == r.tobytes()
