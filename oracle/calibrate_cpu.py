#!/usr/bin/env python3
"""Calibrate the CPU baseline: the REFERENCE itself (eager PyTorch, CPU) against the oracle port, on the same inputs.

TEST INFRASTRUCTURE, build container only (imports /root/reference like gen_golden.py).  bench.py's `cpu_baseline`
times the oracle port on the GPU box, where the reference cannot travel; this script measures, here, how the port
relates to the real thing ("report the here-measured ratio so the proxy is honest").

Path timed: tokens -> tokens_to_bytes -> pull_from_left (data_creation.py:61-67, 179-305) -> embedding gathers -> sum ->
rms-norm (runs/71_mot-in_toks-valemb.py:227-230, 312-314) at config-4 dims (vocab 50257, bpt 16, d 768, byte dim 48).

Run:  python oracle/calibrate_cpu.py [rows] [threads]
"""
from __future__ import annotations

import sys
import time

import numpy as np
import torch

_HERE = __import__("pathlib").Path(__file__).resolve().parent
sys.path.insert(0, str(_HERE.parent))
from oracle import gen_golden as gg  # noqa: E402  (reference loaders; puts tests/ on sys.path)
from oracle import oracle as orc  # noqa: E402
import golden_inputs as gi  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    threads = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    torch.set_num_threads(threads)
    orc.set_threads(threads)
    T, Vt, bpt, D, Db = 2048, 50257, 16, 768, 48
    dc, r71 = gg.load_data_creation(), gg.load_run71_defs()
    tab = gi.widen_left_pad(gi.load_real_ttb8(), bpt)
    toks = gi.fineweb_like_tokens(12345, rows, T, vocab=Vt)
    Et, Eb = gi.normal_table(1, Vt, D).astype(np.float32), gi.normal_table(2, gi.BYTE_VOCAB, Db).astype(np.float32)
    emb = gg.ttb_embedding(tab)
    et, eb = torch.from_numpy(Et), torch.from_numpy(Eb)
    mixin_bytes = r71["mixin_bytes"]

    def reference():
        out = []
        with torch.no_grad():
            for r in range(rows):                       # the run-71 loop body handles one (1, T) row per step
                tk = torch.from_numpy(toks[r:r + 1])
                padded = dc.tokens_to_bytes(tk, emb)
                pulled = dc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
                byte_inputs = pulled.view(T, bpt).t().contiguous()
                out.append(mixin_bytes(et[tk[0].long()][None], eb[byte_inputs].squeeze()))
        return out

    def port():
        padded = orc.tokens_to_bytes(toks, tab.astype(np.float32))
        pulled = orc.pull_from_left(padded, bpt, gi.PAD, gi.EOT)
        return orc.embed_mix(toks, pulled, None, Et, Eb, mode="sum", bpt=bpt, dtype=np.float32, norm_out=True)

    def timeit(fn, budget=8.0):
        fn()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget:
            fn(); n += 1
        return (time.perf_counter() - t0) / n

    a = np.concatenate([x.numpy() for x in reference()], 0)
    b = port()
    print("max |reference - port| =", float(np.abs(a - b).max()))
    tr, tp = timeit(reference), timeit(port)
    n = rows * T
    print(f"reference (torch {torch.__version__} eager, {threads} threads): {n / tr / 1e6:.2f} M tokens/s")
    print(f"oracle port (C/OpenMP, {threads} threads):              {n / tp / 1e6:.2f} M tokens/s")
    print(f"port / reference = {tr / tp:.2f}x")


if __name__ == "__main__":
    main()
