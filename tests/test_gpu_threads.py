"""Two host threads, one context each ("one thread per context; different contexts are independent"): the calls of
both threads are in flight at the same time -- ctypes releases the interpreter lock for the duration of a C call -- on two streams, over
small and large batches, and every result equals what the same context gives alone.  What this guards: process-wide state behind the
C ABI (the per-device "attributes set" flags of the launch helpers, the cached CU count, the packer's error string), which the
reference's one-frame-at-a-time loop (test/metrics_from_model.py:120-300) never exercises."""
import threading

import numpy as np
import pytest
import torch

from conftest import pkg

pytestmark = pytest.mark.gpu


def test_two_threads_two_contexts_give_the_single_thread_bits(calib, gat_weights, mlp_weights):
    syn = pkg('synthetic')
    from conftest import oracle
    sd, prm = gat_weights
    sizes = (1, 7, 33, 120)                                  # latency launches, K-split MLP, tile kernels
    frames = [oracle().processed_input(syn.make_frame(calib, 9100 + i, syn.FrameSpec(persons=1 + i % 5, joint_drop=0.1 * (i % 2)))[0])
              for i in range(max(sizes))]
    engines, streams, batches, want = [], [], [], []
    try:
        for t in range(2):
            eng = pkg('pipeline').Engine(calib.params, calib, max_frames=max(sizes), max_persons_per_camera=6)
            eng.load_gat(sd, prm)
            eng.load_mlp(mlp_weights)
            engines.append(eng)
            streams.append(torch.cuda.Stream(eng.device))
            # thread t walks the sizes in its own order, on its own slice of the frames
            order = sizes if t == 0 else sizes[::-1]
            batches.append([eng.to_device(eng.pack(frames[(3 * t):(3 * t) + n] if (3 * t) + n <= len(frames) else frames[:n])) for n in order])
        for t, eng in enumerate(engines):                    # alone, one after the other
            res = []
            for db in batches[t]:
                sc, pe, npers = eng.match(db)
                po, va = eng.mlp3d(db, pe, npers)
                eng.sync_status()
                res.append([x.cpu().numpy() for x in (sc, pe, npers, po, va)])
            want.append(res)
        errors, barrier = [], threading.Barrier(2)

        def work(t):
            try:
                eng = engines[t]
                with torch.cuda.stream(streams[t]):
                    barrier.wait()
                    for it in range(40):
                        for k, db in enumerate(batches[t]):
                            sc, pe, npers = eng.match(db)
                            po, va = eng.mlp3d(db, pe, npers)
                            eng.sync_status()
                            got = [x.cpu().numpy() for x in (sc, pe, npers, po, va)]
                            w = want[t][k]
                            cnt = w[2]
                            assert np.array_equal(got[0], w[0]) and np.array_equal(got[2], cnt), (t, it, k)
                            for f in range(len(cnt)):
                                c = int(cnt[f])
                                assert np.array_equal(got[1][f, :c], w[1][f, :c]) and np.array_equal(got[4][f, :c], w[4][f, :c]), (t, it, k, f)
                                keep = w[4][f, :c] != 0
                                assert np.array_equal(got[3][f, :c][keep], w[3][f, :c][keep]), (t, it, k, f)
            except BaseException as e:           # noqa: BLE001 -- handed to the main thread
                errors.append((t, repr(e)))
                try:
                    barrier.abort()
                except Exception:
                    pass

        threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=280)
        assert not any(th.is_alive() for th in threads), 'a worker thread did not finish'
        assert not errors, errors
    finally:
        for eng in engines:
            eng.close()
