#!/usr/bin/env python
"""Throughput of the accessory-gene sampler: thin wrapper around `python bench.py --workload genes`
(the benchmark and its CPU-baseline leg live in bench.py).

    python scripts/bench_genes.py [--genes 2000 --samples 32 --haplotypes 6 --vmax 20 --iters 50 --cpu-genes 40]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--haplotypes", type=int, default=6)
    ap.add_argument("--vmax", type=int, default=20)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-genes", type=int, default=40, help="genes of the CPU sample (0 = skip)")
    a = ap.parse_args()
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "genes", "--genes", str(a.genes), "--S", str(a.samples),
           "--G", str(a.haplotypes), "--vmax", str(a.vmax), "--steps", str(a.iters), "--cpu-genes", str(a.cpu_genes)]
    sys.exit(subprocess.call(cmd))
