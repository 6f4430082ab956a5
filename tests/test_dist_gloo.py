"""world_size-2 gloo tests (CPU) of the N>1 path: one flat weight broadcast + round-robin prompt sharding.
No data-path collective exists besides that broadcast."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from layoutllm_t2i_amd import recipe
from layoutllm_t2i_amd.arch import TINY
from layoutllm_t2i_amd.dist import broadcast_packed, checksum, flatten_packed, shard_indices, unflatten_packed
from layoutllm_t2i_amd.weights import pack_state_dict


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_broadcast_and_shard_world2():
    import json
    import subprocess
    world = 2
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err[-2000:]
            line = next(l for l in out.splitlines() if l.startswith("RESULT "))
            res.append(json.loads(line[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res.sort(key=lambda d: d["rank"])
    a, b = res
    assert a["checksum"] == b["checksum"] == a["local_checksum"], "ranks disagree after the weight broadcast"
    assert a["n"] == b["n"] and a["emb_total"] == b["emb_total"] and a["scalars"] == b["scalars"]
    assert sorted(a["shard"] + b["shard"]) == list(range(7)) and not set(a["shard"]) & set(b["shard"])
    assert a["d16"] == b["d16"] == "torch.float16" and a["d32"] == b["d32"] == "torch.float32"


def test_flatten_roundtrip_single_process():
    P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, "cpu", recipe.sd_first_conv(TINY, 0))
    flat, man = flatten_packed(P)
    assert flat.dtype == torch.uint8 and flat.numel() % 256 == 0
    Q = unflatten_packed(flat.clone(), man, TINY, "cpu")
    assert set(Q.w) == set(P.w)
    for k in P.w:
        assert torch.equal(Q.w[k], P.w[k]) and Q.w[k].dtype == P.w[k].dtype, k
        assert Q.w[k].data_ptr() % 16 == 0
    assert Q.s == P.s and Q.emb_offsets == P.emb_offsets and checksum(P) == checksum(Q)


def test_shard_indices_ragged():
    assert shard_indices(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]
    assert shard_indices(5, 7, 8) == []
    assert sum(len(shard_indices(13, r, 4)) for r in range(4)) == 13


def test_bundle_broadcast_world2_one_collective():
    """dist.broadcast_bundle: UNet flat buffer + VAE tensors in ONE tensor broadcast (plus one pickled header); both ranks end
    with identical bytes, dtypes, shapes, the sender's extras and config."""
    import json
    import subprocess
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_bundle_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, worker], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=240)
            assert p.returncode == 0, err[-2000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["checksum"] == b["checksum"] and a["has_sd"] and b["has_sd"] and a["cfg_ok"] and b["cfg_ok"]
    assert a["vae"] == b["vae"] and len(a["vae"]) == 4
    assert a["extra"] == b["extra"] == {"tag": "hello", "n": 3}
    assert a["tensor_broadcasts"] == b["tensor_broadcasts"] == 1


def test_flatten_tensors_roundtrip():
    from layoutllm_t2i_amd.dist import flatten_tensors, unflatten_tensors
    g = torch.Generator().manual_seed(1)
    d = {"b": torch.randn(5, 3, generator=g).half(), "a": torch.randn(7, generator=g), "z": torch.zeros(0)}
    flat, man = flatten_tensors(d, "cpu")
    assert flat.dtype == torch.uint8 and flat.numel() % 256 == 0 and [m[0] for m in man] == ["a", "b", "z"]
    back = unflatten_tensors(flat, man)
    assert all(torch.equal(back[k], d[k]) and back[k].dtype == d[k].dtype for k in d)


@pytest.mark.parametrize("mode", ["src_fails", "rank1_fails"])
def test_sharded_entry_failure_reaches_every_rank_world2(mode):
    """generate_batch_images_sharded: a failure on ONE rank -- src before the conditioning broadcast, or any rank while it
    encodes / denoises its shard -- travels as the payload of the next collective and is raised on the ranks that need it; no
    rank stays blocked in broadcast_object_list / gather_object (the workers would hit the timeout)."""
    import json
    import subprocess
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_fail_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, worker, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=180)
            assert p.returncode == 0, err[-2000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a, b = sorted(res, key=lambda d: d["rank"])
    if mode == "src_fails":
        assert a["got"].startswith("RuntimeError") and b["got"].startswith("RuntimeError") and "rank 0" in a["got"] and "rank 0" in b["got"], (a, b)
    else:
        assert a["got"].startswith("RuntimeError") and "boom on rank 1" in a["got"], a       # src re-raises what rank 1 reported
        assert b["got"].startswith("RuntimeError") and "boom on rank 1" in b["got"], b
