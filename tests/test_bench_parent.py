"""The parent of a multi-GPU bench run (`python bench.py --gpus N` without WORLD_SIZE) must never touch the HIP runtime:
a process that has initialised the GPU may not start other programs on this pool.  It counts the devices from the KFD
topology in sysfs and starts the ranks as fresh processes."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _fake_topology(tmp_path, gpus, cpus=2, render=True, dri_missing=()):
    sysfs = tmp_path / "nodes"
    dri = tmp_path / "dri"
    dri.mkdir()
    node = 0
    for _ in range(cpus):
        d = sysfs / str(node)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        node += 1
    for g in range(gpus):
        d = sysfs / str(node)
        d.mkdir(parents=True)
        minor = 128 + g
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\ngfx_target_version 90500\n")
        if render and g not in dri_missing:
            (dri / f"renderD{minor}").write_text("")
        node += 1
    return str(sysfs), str(dri)


def test_devices_are_counted_from_sysfs(tmp_path):
    import bench
    sysfs, dri = _fake_topology(tmp_path, gpus=8)
    assert bench.visible_gpu_count(sysfs, dri, {}) == 8
    assert bench.visible_gpu_count(sysfs, dri, {"HIP_VISIBLE_DEVICES": "0,1,2"}) == 3
    assert bench.visible_gpu_count(sysfs, dri, {"ROCR_VISIBLE_DEVICES": "4", "HIP_VISIBLE_DEVICES": "0,1"}) == 1
    assert bench.visible_gpu_count(sysfs, dri, {"CUDA_VISIBLE_DEVICES": ""}) == 0
    assert bench.visible_gpu_count(str(tmp_path / "nowhere"), dri, {}) is None


def test_a_container_sees_only_the_gpus_whose_render_nodes_it_was_given(tmp_path):
    import bench
    sysfs, dri = _fake_topology(tmp_path, gpus=8, dri_missing=range(1, 8))
    assert bench.visible_gpu_count(sysfs, dri, {}) == 1
    # no /dev/dri at all: the topology alone decides
    assert bench.visible_gpu_count(sysfs, str(tmp_path / "no_dri"), {}) == 8


def test_the_parent_starts_the_ranks_without_touching_torch_cuda(monkeypatch, tmp_path):
    import torch

    import bench

    def boom(*args, **kwargs):
        raise AssertionError("the parent of a multi-GPU run touched torch.cuda")

    for name in ("device_count", "is_available", "init", "_lazy_init", "set_device", "current_device", "synchronize"):
        monkeypatch.setattr(torch.cuda, name, boom, raising=False)
    for name in ("_cuda_getDeviceCount", "_cuda_init"):
        if hasattr(torch._C, name):
            monkeypatch.setattr(torch._C, name, boom, raising=False)
    sysfs, dri = _fake_topology(tmp_path, gpus=4)
    real_count = bench.visible_gpu_count
    monkeypatch.setattr(bench, "visible_gpu_count", lambda: real_count(sysfs, dri, {}))
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("BENCH_SINGLE_DEVICE", raising=False)
    a = bench.parse(["--gpus", "4", "--workload", "c5"])
    assert bench.launch_ranks(a, ["--gpus", "4", "--workload", "c5"]) == 7        # the ranks' status is the parent's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--master-addr" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--workload", "c5"] and os.path.basename(cmd[-5]) == "bench.py"
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # fewer devices than ranks: refused before anything is started
    seen.clear()
    a8 = bench.parse(["--gpus", "8"])
    assert bench.launch_ranks(a8, ["--gpus", "8"]) == 2 and not seen
    # the one-device rehearsal is let through
    monkeypatch.setenv("BENCH_SINGLE_DEVICE", "1")
    assert bench.launch_ranks(a8, ["--gpus", "8"]) == 7 and "--nproc-per-node=8" in seen["cmd"]


def test_allreduce_every_is_an_option():
    import bench
    assert bench.parse([]).allreduce_every == 0          # one collective, after the last step
    assert bench.parse(["--allreduce-every", "1"]).allreduce_every == 1
