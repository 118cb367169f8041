#!/usr/bin/env python3
"""Does the step slow down over a process's lifetime (seen in tools/ab.py sessions: 16.1 -> 16.9 ms over ~80 s)?  bench.py's step in a loop;
per 100 steps: ms per step, allocator state, host RSS, live Python objects, GPU clocks / temperature (rocm-smi).  A pause of 20 s in the
middle tells heat (recovers) from growth (does not).   usage: python tools/probes/drift.py [config=speaker] [steps=3000]"""
import gc
import os
import re
import resource
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + sys.argv[1:]
config = sys.argv[1] if len(sys.argv) > 1 else "speaker"
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
saved = sys.argv
sys.argv = ["ab.py", config]
from d3net_amd import _lib  # noqa: E402
if os.environ.get("D3_SO"):          # (a variant build of tools/probes/build_variant.sh)
    _lib.SO_PATH = os.path.abspath(os.environ["D3_SO"])
import runpy  # noqa: E402
ns = runpy.run_path(os.path.join(ROOT, "tools", "ab.py"))          # builds model / feeder / step(), runs its 40 settle steps, no switches
step = ns["step"]


def smi():
    try:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showtemp", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
        mclk = re.search(r"mclk clock level: \S+ \((\d+)Mhz\)", o)
        temp = re.search(r"Temperature \(Sensor (?:junction|edge)\) \(C\): ([\d.]+)", o)
        pw = re.search(r"Power \(W\): ([\d.]+)", o)
        return "sclk %s mclk %s temp %s power %s" % tuple(m.group(1) if m else "?" for m in (sclk, mclk, temp, pw))
    except Exception as e:
        return "rocm-smi: %r" % (e,)


print("start:", smi(), flush=True)
t_begin = time.perf_counter()
for i in range(0, nsteps, 100):
    if i == nsteps // 2 // 100 * 100:
        torch.cuda.synchronize()
        print("   -- idle 20 s --", flush=True)
        time.sleep(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 100
    print("steps %5d  t %6.1f s  %.3f ms/step  alloc %.2f GB reserved %.2f GB  rss %.0f MB  objs %d  %s" % (
        i + 100, time.perf_counter() - t_begin, ms, torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30,
        resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024, len(gc.get_objects()), smi() if (i // 100) % 5 == 4 else ""), flush=True)
print("end:", smi())
