#!/usr/bin/env python3
"""GPU box experiment: one kernel on one matrix in one set of allocations, run continuously for a few seconds in blocks of 20 calls,
with the card's clocks / power sampled from sysfs beside it (spgpu_amd/gpu_state.py): is the several-per-cent drift between "fast" and
"slow" runs a state of the card (clocks, power management) rather than of the matrix' placement?
    python tools/exp_clocks.py [rows] [seconds] ; EXP_KIND=powerlaw|uniform ; EXP_IDLE=<s>: an idle pause in the middle"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from spgpu_amd import capi, formats, gpu_state, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
kind = os.environ.get("EXP_KIND", "powerlaw")
card = gpu_state.Card(0)
print("sysfs device:", card.dev, "hwmon:", card.hwmon)
if card.dev:
    print("files:", sorted(os.listdir(card.dev))[:200])
    if card.hwmon:
        print("hwmon files:", sorted(os.listdir(card.hwmon)))
print("identity:", card.identity())
print("idle sample:", card.sample())
handle = capi.create_handle(0)
stream = torch.cuda.Stream()
capi.spgpuSetStream(handle, C.c_void_p(stream.cuda_stream))
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
if kind == "powerlaw":
    lengths = synth.power_law_lengths(n, mean=32.0, max_len=2048, seed=5)
    coo = synth.ragged_coo_on_device(lengths, n, "band", 2048, "D", seed=5)
    h = formats.coo_to_ordered_hell_device(handle, n, *coo, "D", 32, 2048, 256, aligned=True)
    del coo
    r_idx = h["rIdx"]
    alg = h["nnz"] * 12 + n * 12 + n * 8 + (n // 32) * 4 + n * 4
else:
    h = synth.hell_uniform_on_device(n // 32 * 32, 32, "banded", "D", 32, seed=1)
    r_idx = None
    alg = n * 32 * 12 + n * 12 + n * 8 + (n // 32) * 4
x = synth.device_vector(n, "D", 3)
z = torch.zeros(n, dtype=torch.float64, device="cuda")
call = lambda: capi.hellspmv["D"](handle, p(z), None, 1.0, p(h["cM"]), p(h["rP"]), 32, p(h["hack_offsets"]), p(h["rS"]), p(r_idx), 32, n, p(x), 0.0, 0)
with torch.cuda.stream(stream):
    for _ in range(4):
        call()
        stream.synchronize()


def burst(duration, label):
    t_end = time.perf_counter() + duration
    blocks = []
    with gpu_state.Sampler(card, 0.002) as s, torch.cuda.stream(stream):
        while time.perf_counter() < t_end:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(20):
                call()
            b.record(stream)
            b.synchronize()
            blocks.append((time.perf_counter(), a.elapsed_time(b) / 20))
    rows = s.rows
    # per 250 ms: median ms per call, and the clocks sampled in that stretch
    t0 = blocks[0][0]
    edge = 0.25
    for k in range(int(duration / edge) + 1):
        bs = sorted(ms for t, ms in blocks if k * edge <= t - t0 < (k + 1) * edge)
        rs = [r for r in rows if k * edge <= r["t"] - t0 < (k + 1) * edge]
        if not bs:
            continue
        med = lambda key: (sorted(r[key] for r in rs if key in r) or [None])[len([r for r in rs if key in r]) // 2]
        print(f"{label} t={k * edge:5.2f}s  {bs[len(bs) // 2]:.4f} ms/call ({alg / bs[len(bs) // 2] * 1e-6 / 8000:.3f})  min {bs[0]:.4f} max {bs[-1]:.4f}  "
              f"sclk {med('sclk_mhz')} mclk {med('mclk_mhz')} fclk {med('fclk_mhz')} power {med('power_w')} W temp {med('temp_c')} / {med('temp_mem_c')}  ({len(rs)} samples)", flush=True)
    print(f"{label} summary:", s.summary())


burst(seconds, kind)
idle = float(os.environ.get("EXP_IDLE", "0"))
if idle > 0:
    time.sleep(idle)
    print("after an idle pause of", idle, "s:", card.sample())
    burst(seconds, kind + " (after idle)")
