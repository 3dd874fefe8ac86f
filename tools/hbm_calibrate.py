#!/usr/bin/env python3
"""Achievable HBM rates of this GPU for the access shapes of k_gather (calibration for the roofline discussion):
device fill (write only), device copy (read + write), and a read-mostly reduction, 1.1 GB each."""
import time

import torch

n = 1_100_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: a.fill_(65))
print("fill   %.3f ms  %.2f TB/s written" % (1e3 * t, n / t / 1e12))
t = timed(lambda: b.copy_(a))
print("copy   %.3f ms  %.2f TB/s read+written" % (1e3 * t, 2 * n / t / 1e12))
a32 = a.view(torch.int32)
t = timed(lambda: a32.sum())
print("reduce %.3f ms  %.2f TB/s read" % (1e3 * t, n / t / 1e12))
