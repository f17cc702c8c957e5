#!/usr/bin/env python3
"""Decode-step benchmark of the fused SparsePagedFusionAttention op on MI355X.

One "step" = one fused-op call for one layer over the whole batch (B decode tokens):
Q.Cent^T -> top-k -> page selection -> sparse paged attention, one kernel launch.
Default workload = BASELINE.json configs[2] (the configuration the metric is quoted on):
B=64 H=32 (MHA) D=128 seqlen=32k C=2048 k=32, fp16 KV in the vLLM BSH layout, block_size 128,
synthetic inputs in the reference's distributions, a 32 GiB disjoint KV pool (>> the 256 MiB
Infinity Cache) and "budgeted" block_ids so every head selects pageLen/8 = 32 pages (the reference's
design point, fused_ifa_v5_case.py:33-34).

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c4|c5]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts N ranks itself (one process per GPU
under torch.distributed.run, RCCL backend) before it touches the GPU, and exits with their status; started
under torch.distributed.run it is one of the ranks.  Weak scaling: every rank holds `--batch` sequences.

Prints ONE JSON line (rank 0).  `value` = decode tokens/s of the whole job with inputs resident in HBM.
`roofline` prices the fused kernel against the HBM roofline with ALGORITHMIC bytes of the KV gather
(sum over (batch, kv head) of |pages selected by the group| * block_size * D * (sizeof K + sizeof V),
read back from page_position) divided by the kernel's average duration measured with HIP events on the
launch stream.  `cpu_baseline` times the CPU oracle (oracle/oracle.c, OpenMP) on a bounded sample of the
same workload on this box's host cores (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cann-ops-adv-vllm_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (≈6.3 TB/s is what a streaming copy achieves)


# Shape presets (BASELINE.json configs[2..4]); 64 sequences per GPU in all of them.
WORKLOADS = {
    "c3": dict(batch=64, heads=32, kv_heads=32, head_dim=128, seqlen=32 * 1024, clusters=2048, topk=32, kv="f16"),
    "c4": dict(batch=64, heads=32, kv_heads=32, head_dim=128, seqlen=64 * 1024, clusters=2048, topk=32, kv="int8"),
    "c5": dict(batch=64, heads=32, kv_heads=8, head_dim=128, seqlen=128 * 1024, clusters=2048, topk=64, kv="f16"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=50)   # the first ~50 launches of a process average 4 us more than the rest
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU")
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--kv-heads", type=int, default=32)
    ap.add_argument("--head-dim", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=32 * 1024)
    ap.add_argument("--clusters", type=int, default=2048)
    ap.add_argument("--topk", type=int, default=32)
    ap.add_argument("--block-size", type=int, default=128)
    ap.add_argument("--kv", choices=["f16", "bf16", "int8"], default="f16")
    ap.add_argument("--kv-layout", choices=["bsh", "bnsd"], default="bsh",
                    help="bsh = [blocks, block_size, Hkv, D] (the reference's layout, the default); bnsd = [blocks, Hkv, "
                         "block_size, D] (a head's rows of a page are contiguous)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-tail-sharing", action="store_true", help="diagnostic: spfaSetTailSharing(0)")
    ap.add_argument("--force-tail-sharing", action="store_true", help="diagnostic: spfaSetTailSharing(2)")
    ap.add_argument("--no-shared-scoring", action="store_true", help="diagnostic: spfaSetSharedScoring(0)")
    ap.add_argument("--allreduce-every", type=int, default=0,
                    help="multi-GPU: all-reduce the global max_page_position_length every N-th step; 0 (default): once, "
                         "after the last timed step (the final max_page_position_length), inside the timed region")
    ap.add_argument("--no-dropin", action="store_true",
                    help="skip the extra leg that times the reference-shaped call (custom_ops.sparse_paged_fusion_attention: "
                         "new descriptors, phase 1 and a new workspace on every call) beside the prepared launch")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default=None,
                    help="preset shape: c3 = BASELINE configs[2] (the default shape), c4 = configs[3] (int8 KV, S=64k), "
                         "c5 = configs[4] (GQA 32/8, S=128k, k=64; 64 sequences per GPU, 512 in all on 8 GPUs)")
    a = ap.parse_args(argv)
    if a.workload:   # the preset supplies the defaults; options given beside it (--batch 8) still count
        ap.set_defaults(**WORKLOADS[a.workload])
        a = ap.parse_args(argv)
    return a


def build_workload(a, device, B, seed, custom_ops):
    """Synthetic decode state for B sequences on `device` (all tensors resident in HBM)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    Hq, Hkv, D, C, k, bs = a.heads, a.kv_heads, a.head_dim, a.clusters, a.topk, a.block_size
    G = Hq // Hkv
    page_len = (a.seqlen + bs - 1) // bs
    n_pages = B * page_len
    qdt = torch.bfloat16 if a.kv == "bf16" else torch.float16
    eps = 1e-3
    kv_shape = (n_pages, bs, Hkv, D) if getattr(a, "kv_layout", "bsh") == "bsh" else (n_pages, Hkv, bs, D)
    q = (torch.rand((B, Hq, D), generator=g, device=device) * (1 - eps) + eps).to(qdt)
    l1 = torch.randn((Hkv, C, D), generator=g, device=device).to(qdt)
    if a.kv == "int8":
        key = torch.empty(kv_shape, dtype=torch.int8, device=device)
        val = torch.empty_like(key)
        step = max(1, (1 << 30) // (bs * Hkv * D))
        for i in range(0, n_pages, step):
            key[i:i + step] = torch.randint(-127, 128, key[i:i + step].shape, generator=g, device=device, dtype=torch.int8)
            val[i:i + step] = torch.randint(-127, 128, val[i:i + step].shape, generator=g, device=device, dtype=torch.int8)
        a_scale = (torch.rand((2, Hkv, D), generator=g, device=device) * 0.02 + 0.001).to(qdt)
        a_off = torch.randn((2, Hkv, D), generator=g, device=device).to(qdt)
    else:
        key = torch.empty(kv_shape, dtype=qdt, device=device)
        val = torch.empty_like(key)
        step = max(1, (1 << 30) // (bs * Hkv * D * 2))
        for i in range(0, n_pages, step):
            key[i:i + step].uniform_(eps, 1.0, generator=g)
            val[i:i + step].uniform_(eps, 1.0, generator=g)
        a_scale = a_off = None
    # disjoint block table: every (sequence, logical page) owns one physical page
    table = torch.randperm(n_pages, generator=g, device=device).to(torch.int32).reshape(B, page_len)
    seq = torch.full((B,), a.seqlen, dtype=torch.int32, device=device)
    # budgeted block_ids: per (sequence, kv head) exactly page_len/8 pages carry a cluster id from the
    # true top-k of one of the group's q heads, every other page an id outside the group's top-k sets
    topk = custom_ops.compute_cent(q, l1, k=k).long()                       # [B,Hq,k]
    member = torch.zeros((B, Hkv, C), dtype=torch.bool, device=device)
    member.scatter_(2, topk.reshape(B, Hkv, G * k), True)
    ids = torch.randint(0, C, (B, Hkv, page_len), generator=g, device=device)
    for _ in range(64):
        bad = member.gather(2, ids)
        if not bad.any():
            break
        ids = torch.where(bad, torch.randint(0, C, ids.shape, generator=g, device=device), ids)
    n_sel = max(1, page_len // 8)
    sel = torch.rand((B, Hkv, page_len), generator=g, device=device).argsort(dim=2)[:, :, :n_sel]   # pages
    head_of = sel % G                                                        # which q head of the group
    pick = torch.randint(0, k, sel.shape, generator=g, device=device)
    tk = topk.reshape(B, Hkv, G, k)
    chosen = tk.gather(2, head_of.unsqueeze(-1).expand(-1, -1, -1, k)).gather(3, pick.unsqueeze(-1)).squeeze(-1)
    ids.scatter_(2, sel, chosen)
    block_ids = torch.zeros((Hkv, n_pages), dtype=torch.int32, device=device)
    phys = table.long().unsqueeze(1).expand(B, Hkv, page_len)
    block_ids.scatter_(1, phys.permute(1, 0, 2).reshape(Hkv, -1), ids.permute(1, 0, 2).reshape(Hkv, -1).to(torch.int32))
    return dict(q=q, l1=l1, key=key, val=val, table=table, seq=seq, block_ids=block_ids, a_scale=a_scale,
                a_off=a_off, page_len=page_len, n_pages=n_pages)


def workload_name(a):
    base = "sparse_paged_fusion_attention decode step"
    shape = (a.batch, a.heads, a.kv_heads, a.head_dim, a.seqlen, a.clusters, a.topk, a.block_size, a.kv)
    if shape == (64, 32, 32, 128, 32768, 2048, 32, 128, "f16"):
        return base + " (BASELINE configs[2])"
    if shape == (64, 32, 32, 128, 65536, 2048, 32, 128, "int8"):
        return base + " (BASELINE configs[3], int8 KV)"
    if shape == (64, 32, 8, 128, 131072, 2048, 64, 128, "f16"):
        return base + " (BASELINE configs[4], one GPU's 64 of the 512 sequences)"
    return base + " (custom shape)"


def gather_bytes(pos, plen, Hkv, G, max_page, bs, D, kv_elem):
    """Algorithmic KV bytes of one launch: pages shared inside a GQA group count once per kv head."""
    B, Hq, P = pos.shape
    valid = torch.arange(P, device=pos.device)[None, None, :] < plen[:, :, :1]
    hit = torch.zeros((B, Hkv, max_page + 1), dtype=torch.bool, device=pos.device)
    idx = torch.where(valid, pos.long(), torch.full_like(pos, max_page).long()).reshape(B, Hkv, G * P)
    hit.scatter_(2, idx, True)
    pages = int(hit[:, :, :max_page].sum().item())
    return pages, pages * bs * D * kv_elem * 2


def host_sample(a, wl, seqs):
    """Copy the decode state of the sequences `seqs` to the host (their KV pages compacted into a small pool, so
    only len(seqs)*page_len pages cross PCIe) and return a callable that runs the CPU oracle on it."""
    from oracle import cport
    seqs = torch.as_tensor(seqs, dtype=torch.long, device=wl["q"].device)
    Bs = int(seqs.numel())
    page_len = wl["page_len"]
    phys = wl["table"][seqs].reshape(-1).long()
    as16 = lambda t: t.cpu().view(torch.int16).numpy().view(np.uint16)
    key = wl["key"][phys].cpu().numpy() if a.kv == "int8" else as16(wl["key"][phys])
    val = wl["val"][phys].cpu().numpy() if a.kv == "int8" else as16(wl["val"][phys])
    table = np.arange(Bs * page_len, dtype=np.int32).reshape(Bs, page_len)
    ids = wl["block_ids"][:, phys].cpu().numpy()
    dt = "bf16" if a.kv == "bf16" else "f16"
    kw = dict(kv_dt="i8" if a.kv == "int8" else dt, dt=dt)
    if a.kv == "int8":
        kw.update(antiq_scale=as16(wl["a_scale"]), antiq_offset=as16(wl["a_off"]))
    q, l1 = as16(wl["q"][seqs]), as16(wl["l1"])
    seq = wl["seq"][seqs].cpu().numpy()
    scale = a.head_dim ** -0.5
    return lambda: cport.fusion(q, key, val, table, l1, ids, seq, a.heads, a.kv_heads, scale, a.block_size, a.topk,
                                256, **kw)


def cpu_baseline(a, wl, custom_ops, plan_args):
    """Time the CPU oracle (C/OpenMP port) on the first `cpu_sample_batch` sequences of the workload."""
    from oracle import cport
    Bs = min(a.cpu_sample_batch, wl["q"].shape[0])
    run = host_sample(a, wl, list(range(Bs)))
    out, pos, len8, max8 = run()
    # the sample doubles as a full-size-shape parity check of the GPU result; the GPU output carries one
    # rounding to its dtype (half an ulp: 2^-11 relative for f16, 2^-8 for bf16) on top of the fp32 error
    got = plan_args["out"][:Bs].float().cpu().numpy()
    gpos = plan_args["pos"][:Bs].cpu().numpy()
    denom = np.maximum(np.abs(out), 1e-2 * np.abs(out).max())
    parity = {"page_position_equal": bool(np.array_equal(gpos, pos)),
              "attention_max_rel_err": float((np.abs(got - out) / denom).max()),
              "rel_err_denominator": "max(|want|, 1e-2 * max|want|): elements below 1 % of the largest output are "
                                     "measured against that floor",
              "output_rounding_half_ulp": 2.0 ** (-8 if a.kv == "bf16" else -11)}
    t0 = time.perf_counter()
    reps = 0
    while True:
        run()
        reps += 1
        el = time.perf_counter() - t0
        if el > 12.0 or reps >= 5000:
            break
    return {"value": Bs * reps / el, "unit": "tokens/s", "cores": cport.threads(), "kind": "port",
            "sample": f"first {Bs} of {wl['q'].shape[0]} sequences (same H/D/S/C/k, KV pages compacted to the host), "
                      f"{reps} reps in {el:.1f}s of oracle/oracle.c (OpenMP)",
            "ms_per_step_sample": 1e3 * el / reps, "parity_vs_gpu": parity}


def kernel_source_stamp():
    """Short sha256 over the kernel and planner sources: ties a committed HBM-traffic measurement to the code it was
    taken on, so a number measured on an older kernel is never reported for a newer one."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "cann-ops-adv-vllm_amd", "csrc")
    for name in sorted(os.listdir(src)):
        if name.endswith((".hip", ".inc", ".h", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(src, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def measured_traffic(key):
    """(HBM bytes per launch, note) from profiles/traffic.json (rocprofv3 --pmc passes, tools/profile_round.sh); the
    bytes are None when there is no entry for this workload or it was measured on other kernel sources."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, "no profiles/traffic.json"
    try:
        with open(tfile) as f:
            rec = json.load(f)
    except Exception as e:
        return None, f"profiles/traffic.json unreadable: {e!r}"
    ent = rec.get(key)
    if ent is None:
        return None, "no PMC measurement committed for this workload"
    stamp = kernel_source_stamp()
    if ent.get("kernel_source_stamp") != stamp:
        return None, (f"stale: measured on kernel sources {ent.get('kernel_source_stamp')}, this tree is {stamp} "
                      "(re-run tools/profile_round.sh)")
    return ent.get("hbm_bytes_per_launch"), "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, separate passes, same sources"


def visible_gpu_count(sysfs="/sys/class/kfd/kfd/topology/nodes", dri="/dev/dri", environ=None):
    """GPUs this process could use, found WITHOUT the HIP runtime (a process that has initialised the GPU must not
    start the ranks): the KFD topology in sysfs lists one node per agent, GPUs are the nodes with simd_count > 0, and in a
    container a GPU counts only when its render node is present under /dev/dri.  HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the count like they narrow the runtime's view.  Returns None when
    the topology cannot be read (the ranks then find out for themselves)."""
    environ = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(sysfs), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    have_dri = os.path.isdir(dri)
    n = 0
    for node in nodes:
        props = {}
        try:
            with open(os.path.join(sysfs, node, "properties")) as f:
                for line in f:
                    parts = line.split()
                    if len(parts) == 2:
                        props[parts[0]] = parts[1]
        except OSError:
            continue
        try:
            if int(props.get("simd_count", "0")) <= 0:
                continue
            minor = int(props.get("drm_render_minor", "-1"))
        except ValueError:
            continue
        if have_dri and minor >= 0 and not os.path.exists(os.path.join(dri, f"renderD{minor}")):
            continue
        n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        val = environ.get(var)
        if val is not None:
            listed = [x for x in val.split(",") if x.strip() != ""]
            n = min(n, len(listed))
    return n


def launch_ranks(a, argv=None):
    """Parent of a multi-GPU run started as plain `python bench.py --gpus N`: start one rank per GPU under
    torch.distributed.run and return their exit status.  This process never touches the HIP runtime — the devices are
    counted from sysfs — so nothing that has initialised the GPU is ever replaced or forked; the ranks are fresh
    processes, and any rank that fails makes the whole run fail."""
    import socket
    import subprocess
    n_dev = visible_gpu_count()
    if n_dev is not None and n_dev < a.gpus and os.environ.get("BENCH_SINGLE_DEVICE") != "1":
        print(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible GPUs but this machine shows {n_dev}; "
              "nothing was run (BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo rehearses the ranks on one device)",
              file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
        (sys.argv[1:] if argv is None else list(argv))
    return subprocess.call(cmd, env=env)


def dropin_leg(a, wl, custom_ops, _abi, pos, plen, pmax, prepared_out):
    """The reference-shaped call beside the prepared launch: custom_ops.sparse_paged_fusion_attention with the
    reference's positional arguments — descriptors, phase 1 (validation + planning), a workspace from the allocator,
    phase 2 and the destruction of everything, on every call, as extension/csrc/add_custom.cpp:770-935 does.
    ms_per_step: the K calls between two events on the launch stream; host_ms_per_call: what the calling thread spends
    per call (it only enqueues); kernels_per_call: from the library's own launch count."""
    dev = wl["q"].device
    e = torch.empty(0, dtype=wl["q"].dtype, device=dev)
    e64 = torch.empty(0, dtype=torch.int64, device=dev)
    hint = [max(1, a.seqlen // 8)]
    scale = a.head_dim ** -0.5

    def call():
        return custom_ops.sparse_paged_fusion_attention(
            wl["q"], wl["key"], wl["val"], e, e, hint, e, e, e, e, e,
            wl["a_scale"] if wl["a_scale"] is not None else e, wl["a_off"] if wl["a_off"] is not None else e,
            wl["table"], e64, wl["l1"], wl["block_ids"], wl["seq"], pos, plen, pmax,
            a.heads, scale, "BNSD", a.kv_heads, a.block_size, 1, k=a.topk)[0]

    torch.cuda.synchronize()
    pos_prepared = pos.clone()
    for _ in range(max(5, a.warmup // 5)):
        out = call()
    torch.cuda.synchronize()
    same = bool(torch.equal(pos, pos_prepared)) and float((out.float() - prepared_out.float()).abs().max()) <= \
        2e-3 * float(prepared_out.float().abs().max())
    n0 = int(_abi.spfaKernelLaunches())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        call()
    host = time.perf_counter() - t0
    ev1.record()
    torch.cuda.synchronize()
    n1 = int(_abi.spfaKernelLaunches())
    return {"ms_per_step": ev0.elapsed_time(ev1) / a.steps, "host_ms_per_call": 1e3 * host / a.steps,
            "kernels_per_call": (n1 - n0) / a.steps, "agrees_with_prepared": same,
            "what": "custom_ops.sparse_paged_fusion_attention(...) with the reference's positional arguments: new "
                    "descriptors, phase 1, a new workspace and phase 2 on every call"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    if os.environ.get("BENCH_WATCHDOG"):   # rehearsal aid: dump every thread's stack and exit if the run stalls
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["BENCH_WATCHDOG"]), exit=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; reporting the {world} rank(s) that exist", file=sys.stderr)
    dist = None
    # Rehearsal on a one-GPU box: BENCH_SINGLE_DEVICE=1 puts every rank on cuda:0 and BENCH_DIST_BACKEND=gloo replaces
    # RCCL (which refuses two ranks on one device); the driver's multi-GPU runs use neither.
    if os.environ.get("BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    backend = "none"
    # BENCH_FORCE_DIST=1 (rehearsal aid): a one-rank process group, so that RCCL and the side-stream collective run on a
    # box with a single GPU too
    force_dist = world == 1 and os.environ.get("BENCH_FORCE_DIST") == "1"
    if force_dist:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries exactly one JSON line,
        # so file descriptor 1 points at stderr until the communicator exists
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.all_reduce(torch.zeros(1, device=torch.device("cuda", local_rank)))
                torch.cuda.synchronize()
            else:
                dist.init_process_group(backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    import custom_ops
    from custom_ops import _abi, shard

    if os.environ.get("SPFA_GEOM"):      # experiments: force the workgroup geometry of the one-q-head kernels (1 wide, 2 narrow)
        _abi.spfaSetWorkgroupGeometry(int(os.environ["SPFA_GEOM"]))
    if os.environ.get("SPFA_BATCHED_SELECTION"):
        _abi.spfaSetBatchedSelection(int(os.environ["SPFA_BATCHED_SELECTION"]))
    if os.environ.get("SPFA_K_THROUGH_LDS"):
        _abi.spfaSetKThroughLds(int(os.environ["SPFA_K_THROUGH_LDS"]))
    if a.no_tail_sharing:
        _abi.spfaSetTailSharing(0)
    if a.force_tail_sharing:
        _abi.spfaSetTailSharing(2)
    if a.no_shared_scoring:
        _abi.spfaSetSharedScoring(0)
    B = a.batch                                     # weak scaling: `batch` sequences per GPU
    wl = build_workload(a, device, B, a.seed + 1000 * rank, custom_ops)
    Hq, Hkv, D, bs = a.heads, a.kv_heads, a.head_dim, a.block_size
    pos = torch.empty((B, Hq, 256), dtype=torch.int32, device=device)
    plen = torch.empty((B, Hq, 8), dtype=torch.int32, device=device)
    pmax = torch.empty((B, 8), dtype=torch.int64, device=device)
    def make_plan(pos_, plen_, pmax_):
        return custom_ops.PreparedFusion(wl["q"], wl["key"], wl["val"], wl["table"], wl["l1"], wl["block_ids"], wl["seq"],
                                         pos_, plen_, pmax_, Hq, D ** -0.5, Hkv, bs, antiquant_scale=wl["a_scale"],
                                         antiquant_offset=wl["a_off"], k=a.topk)

    plan = make_plan(pos, plen, pmax)
    if os.environ.get("SPFA_DESCRIBE") and rank == 0:
        print("[bench] plan:", plan.describe(), file=sys.stderr)
    reducer = shard.GlobalMaxLength(device, every=a.allreduce_every,
                                    depth=int(os.environ.get("BENCH_REDUCER_DEPTH", "4"))) if dist is not None else None
    # With a collective behind every launch the decode loop alternates two sets of output tensors (as a serving loop
    # that hands step i's outputs on while step i + 1 runs would): a launch then never waits for the snapshot of the
    # launch just before it.  BENCH_SINGLE_OUTPUTS=1 keeps one set (the A/B of that choice).
    plans = [(plan, pmax)]
    if reducer is not None and os.environ.get("BENCH_SINGLE_OUTPUTS") != "1":
        pos2, plen2, pmax2 = torch.empty_like(pos), torch.empty_like(plen), torch.empty_like(pmax)
        plans.append((make_plan(pos2, plen2, pmax2), pmax2))
    step_no = [0]

    def step():
        pl, pm = plans[step_no[0] % len(plans)]
        step_no[0] += 1
        if reducer is not None:
            reducer.before_launch(pm)
        pl.run()
        if reducer is not None:      # the only collective of the path: global max_page_position_length
            reducer.submit(pm)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # The timed region: exactly K steps between two events on the launch stream (torch's current stream is the stream
    # the library enqueues on); their span / K is the kernel time of a launch.
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for i in range(a.steps):
        step()
    ev1.record()
    if reducer is not None and a.allreduce_every <= 0:
        # the final max_page_position_length over all ranks: the one collective the sharded path needs
        reducer.every = 1
        reducer.submit(plans[(step_no[0] - 1) % len(plans)][1])
        reducer.every = 0
    torch.cuda.synchronize()
    if reducer is not None:
        reducer.wait()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / a.steps
    # The spread of single launches comes from a second pass of K steps with an event in front of every step and one
    # behind the last.  It is kept out of the timed region because an event between two launches keeps the second from
    # being set up while the first still runs (measured: about 3.5 us per launch, tools/event_overhead.py).
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    for i in range(a.steps):
        evs[i].record()
        step()
    evs[a.steps].record()
    torch.cuda.synchronize()
    if reducer is not None:
        reducer.wait()
    per_step = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(a.steps)])
    rank_kernel_ms = [kernel_ms]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([kernel_ms], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_kernel_ms = [float(x.item()) for x in every]
    kv_elem = 1 if a.kv == "int8" else 2
    pages, gbytes = gather_bytes(pos, plen, Hkv, Hq // Hkv, wl["page_len"], bs, D, kv_elem)
    achieved = gbytes / (kernel_ms * 1e-3) / 1e9
    traffic, traffic_note = measured_traffic(f"B{B}_H{Hq}_KV{Hkv}_D{D}_S{a.seqlen}_C{a.clusters}_k{a.topk}_{a.kv}")
    tail = "off (attention_out bit-reproducible)" if a.no_tail_sharing else \
        "planner default: on for this launch shape when a workgroup has several items (attention_out then agrees " \
        "between launches to fp32 rounding, <= 2e-6 relative, not bit for bit; integer outputs always bit-identical)"
    result = {
        "metric": "decode tokens/sec + KV HBM GB/s vs roofline, B=64 H=32 D=128 S=32k k=32",
        "value": world * B * a.steps / elapsed, "unit": "tokens/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": {"f16": "f16", "bf16": "bf16", "int8": "i8"}[a.kv], "data": "synthetic",
        "config": {"workload": workload_name(a),
                   "batch_per_gpu": B, "global_batch": world * B, "heads": Hq, "kv_heads": Hkv, "head_dim": D,
                   "seq_len": a.seqlen, "clusters": a.clusters, "topk": a.topk, "block_size": bs, "kv_dtype": a.kv,
                   "kv_layout": ("BSH [blocks, block_size, Hkv, D]" if a.kv_layout == "bsh" else "BNSD [blocks, Hkv, block_size, D]"), "kv_pool_GiB": round(2 * wl["key"].numel() * kv_elem / 2 ** 30, 2),
                   "pages_selected_per_launch": pages, "parallelism": f"batch-shard x{world}",
                   "tail_sharing": tail},
        "ranks": {"world_size": world, "backend": {"nccl": "rccl"}.get(backend, backend),
                  "ranks_seen": (dist.get_world_size() if dist is not None else 1),
                  "kernel_ms_per_rank": rank_kernel_ms,
                  "collective": ((f"all_reduce(MAX) of one int64 every {a.allreduce_every} step(s) on a side stream"
                                  if a.allreduce_every > 0 else
                                  "one all_reduce(MAX) of one int64 after the last timed step (the final "
                                  "max_page_position_length), inside the timed region") if dist is not None else "none")},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_note": traffic_note,
                     "kernel_ms": kernel_ms,
                     "kernel_ms_stats": {"min": float(per_step.min()), "median": float(np.median(per_step)),
                                         "p95": float(np.percentile(per_step, 95)), "max": float(per_step.max()),
                                         "n": int(a.steps),
                                         "note": "second pass of the same K launches with an event between every two "
                                                 "(the events cost about 3.5 us per launch, so kernel_ms, from the "
                                                 "two events around the timed region, is lower than this median)"},
                     "algorithmic_bytes_per_launch": gbytes, "scope": "one GPU (rank 0)"},
    }
    if reducer is not None:
        result["ranks"]["allreduce_every"] = a.allreduce_every
        result["ranks"]["collectives_submitted"] = reducer.submitted
        result["ranks"]["collective_ms"] = reducer.collective_ms()   # rank 0's, events on the collective's stream
    if rank == 0 and world == 1 and not a.no_dropin and not force_dist:
        try:
            result["dropin"] = dropin_leg(a, wl, custom_ops, _abi, pos, plen, pmax, plan.out.clone())
            result["dropin"]["vs_prepared"] = result["dropin"]["ms_per_step"] / kernel_ms
        except Exception as e:
            result["dropin"] = {"ms_per_step": None, "error": repr(e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            result["cpu_baseline"] = cpu_baseline(a, wl, custom_ops, {"out": plan.out, "pos": pos})
        except Exception as e:   # the GPU number stands on its own; say why the CPU leg is missing
            result["cpu_baseline"] = {"value": None, "error": repr(e)}
    if reducer is not None:
        result["ranks"]["output_sets"] = len(plans)
    if rank == 0:
        print(json.dumps(result), flush=True)
    for pl, _ in plans:
        pl.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
