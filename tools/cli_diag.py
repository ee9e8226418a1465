#!/usr/bin/env python3
"""GPU-box diagnostic: the smart-pairing scenario of tests/test_cli.py::test_cli_gpu, with and without device CIGARs."""
import os, subprocess, sys, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import testdata, test_cli
from bwa_amd import build as b
_, cli = b.build_host(verbose=False)
fa, g = testdata.medium_index()
tmp = pathlib.Path("/tmp/clidiag"); tmp.mkdir(exist_ok=True)
f1, f2, inter, fasta = test_cli._write_inputs(tmp, g, 20000, seed=402)
x = ["-p", "-C", "-R", "@RG\\tID:rg1\\tSM:s"]
for cig in sys.argv[1:] or ["0", "1"]:
    env = dict(os.environ, BWAGPU_CLI_CIGARS=cig, BWAGPU_CLI_WATCHDOG="10", BWAGPU_CLI_TRACE="1")
    p = subprocess.run([cli, "mem", "-K", "100000000", "-t", "4", "-v", "1"] + x + [fa, inter], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
    print(f"cigars={cig}: rc={p.returncode} out={len(p.stdout)} bytes\n" + p.stderr.decode()[-700:], flush=True)
