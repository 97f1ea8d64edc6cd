"""Where the time of the two-handle submit / wait pipeline goes: host time inside submit and inside wait, per call, against the synchronous
call.  Usage on the GPU box: python3 tools/time_submit_wait.py [frames per call]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
W, H = 640, 480
fr = pkg.host_array((B, H, W), np.uint8)
for i in range(B):
    fr[i] = pkg.synth_frame(i % 64, W, H)
exs = [pkg.ORBextractor(1000, 1.2, 8, 20, 7, max_width=W, max_height=H, max_batch=B) for _ in range(2)]
outs = [e.output_arrays(B, page_locked=True) for e in exs]
for e, o in zip(exs, outs):
    e.extract_batch(fr, (0, 1000), out=o)
t = time.perf_counter()
for _ in range(6):
    exs[0].extract_batch(fr, (0, 1000), out=outs[0])
sync_ms = (time.perf_counter() - t) / 6 * 1e3
print(f"synchronous call: {sync_ms:.2f} ms per {B} frames = {B / sync_ms * 1e3:.0f} frames/s")
for depth in (1, 2):
    for lanes in (1, 2):
        ts, tw = [], []
        oo = [[e.output_arrays(B, page_locked=True) for _ in range(depth)] for e in exs[:lanes]]
        t0 = time.perf_counter()
        for k in range(lanes):
            for d in range(depth):
                a = time.perf_counter(); exs[k].extract_batch_submit(fr, (0, 1000), out=oo[k][d]); ts.append(time.perf_counter() - a)
        reps = 8
        for r in range(reps - 1):
            for d in range(depth):
                for k in range(lanes):
                    a = time.perf_counter(); exs[k].extract_batch_wait(); tw.append(time.perf_counter() - a)
                    a = time.perf_counter(); exs[k].extract_batch_submit(fr, (0, 1000), out=oo[k][d]); ts.append(time.perf_counter() - a)
        for k in range(lanes):
            for d in range(depth):
                a = time.perf_counter(); exs[k].extract_batch_wait(); tw.append(time.perf_counter() - a)
        wall = time.perf_counter() - t0
        print(f"{lanes} handle(s) x {depth} batch(es) in flight: {lanes * depth * reps * B / wall:.0f} frames/s; host time in submit {np.mean(ts) * 1e3:.2f} ms "
              f"(max {np.max(ts) * 1e3:.2f}), in wait {np.mean(tw) * 1e3:.2f} ms")
