"""The exec phase only enqueues (a memset of the workspace counters and one kernel): a prepared launch can be
captured into a HIP graph and replayed, which is how a serving loop would drive it.  The reference's binding
re-plans on every call (extension/csrc/add_custom.cpp:836-895); its executor can be marked repeatable (:847)."""
import numpy as np
import pytest
import torch

import datagen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Hq,Hkv,B", [(8, 8, 3), (8, 2, 3), (32, 32, 40)])
def test_prepared_launch_in_a_hip_graph(Hq, Hkv, B):
    import custom_ops
    import gpu_helpers as H
    D, C, topk, bs, max_page, kv_page_len = 128, 256, 8, 128, 40, 120
    seq_lens = [int(x) for x in datagen.rng_for(1).integers(1, max_page * bs, size=B)]
    q, l1 = datagen.cent_case(11, B, Hq, Hkv, D, C, grid=True)
    ids, table, seq = datagen.page_case(11, Hkv, C, kv_page_len, B, max_page, seq_lens)
    k, v = datagen.kv_case(11, kv_page_len, bs, Hkv, D, "BSH", dist="randn")
    qd, kd, vd, l1d, td, idsd, seqd = (H.dev(x) for x in (q, k, v, l1, table, ids, seq))
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=H.DEV)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=H.DEV)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=H.DEV)
    plan = custom_ops.PreparedFusion(qd, kd, vd, td, l1d, idsd, seqd, pos, plen, pmax, Hq, D ** -0.5, Hkv, bs, k=topk,
                                     out_dtype=torch.float32)
    plan.run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        plan.run()                                   # warm-up on the capture stream
        side.synchronize()
        with torch.cuda.graph(graph, stream=side):
            plan.run()
    for step in range(3):
        q2, _ = datagen.cent_case(100 + step, B, Hq, Hkv, D, C, grid=True)
        qd.copy_(H.dev(q2))                          # tensors are captured by pointer: update in place
        seqd.copy_(H.dev(np.maximum(1, seq - step * 17).astype(np.int32)))
        plan.out.zero_(); pos.zero_(); plen.zero_(); pmax.zero_()
        graph.replay()
        torch.cuda.synchronize()
        got = [t.clone() for t in (plan.out, pos, plen, pmax)]
        plan.run()                                   # the same launch, eagerly
        torch.cuda.synchronize()
        assert torch.equal(got[1], pos) and torch.equal(got[2], plen) and torch.equal(got[3], pmax)
        scale = float(plan.out.abs().max())
        assert float((got[0] - plan.out).abs().max()) <= 2e-6 * scale
        assert bool(torch.isfinite(got[0]).all()) and float(got[0].abs().max()) > 0
    plan.close()
