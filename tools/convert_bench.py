#!/usr/bin/env python3
"""convertFromHNSW at size: the CPU harness (all host threads) beside the GPU path, byte comparison of the two files.
Usage: convert_bench.py <vanilla index file> <dim>"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hsutil import load_product
hs = load_product()
hp, dim = sys.argv[1], int(sys.argv[2])
thr = min(len(os.sched_getaffinity(0)), 64)
a, b = "/tmp/conv_cpu.slim", "/tmp/conv_gpu.slim"
t0 = time.time(); hs.convert_slim(hp, a, dim, threads=thr); t_cpu = time.time() - t0
t0 = time.time(); hs.convert_slim(hp, a + "1", dim, threads=1); t_cpu1 = time.time() - t0
hs.convert_slim_gpu(hp, b, dim, threads=thr)   # warm-up (module load, allocations)
t0 = time.time(); used, ms = hs.convert_slim_gpu(hp, b, dim, threads=thr); t_gpu = time.time() - t0
same = open(a, "rb").read() == open(b, "rb").read() and open(a + "1", "rb").read() == open(b, "rb").read()
n = os.path.getsize(hp)
print(f"convertFromHNSW of {hp} ({n / 1e6:.0f} MB): CPU harness {thr} threads {t_cpu:.2f} s (load + convert + save), 1 thread {t_cpu1:.2f} s; "
      f"GPU path {t_gpu:.2f} s end to end (load + upload + kernels {ms:.1f} ms + host assembly on {thr} threads + save), used_gpu={used}; files byte-identical: {same}")
