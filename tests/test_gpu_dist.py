"""Multi-GPU correctness without a multi-GPU node (multi-GPU row): two ranks over gloo on ONE GPU.
A rank's output for a given (prompt, seed) must equal, BITWISE, what a single process produces for the same samples:
the weight broadcast is lossless, noise / conditioning are per-prompt (not per-rank), shards are disjoint and complete,
and nothing in the denoising path depends on the rank."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rank_outputs_equal_single_process_bitwise(tmp_path):
    import dist_gpu_worker as W
    from layoutllm_t2i_amd import recipe
    from layoutllm_t2i_amd.arch import TINY
    from layoutllm_t2i_amd.dist import shard_indices
    from layoutllm_t2i_amd.weights import pack_state_dict
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, worker, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-3000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res.sort(key=lambda d: d["rank"])
    assert res[0]["checksum"] == res[1]["checksum"] and res[0]["has_sd"] and res[1]["has_sd"]
    assert sorted(res[0]["shard"] + res[1]["shard"]) == list(range(W.N_PROMPTS))
    # the same shards, in THIS process, from weights packed here (no broadcast)
    dev = "cuda:0"
    P = pack_state_dict(recipe.state_dict(TINY, 0), TINY, torch.device(dev), recipe.sd_first_conv(TINY, 0))
    model = W.model_from_packed(P, dev)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npz")
        mine = shard_indices(W.N_PROMPTS, r, world)
        assert list(got["idx"]) == mine
        ref = W.run_shard(model, mine, dev).numpy()
        assert np.isfinite(ref).all() and np.array_equal(got["lat"], ref), f"rank {r}: max diff {np.abs(got['lat'] - ref).max()}"
    # and a sample's latent does not depend on its batch neighbours beyond dispatch-dependent fp16 rounding
    solo = W.run_shard(model, [2], dev).numpy()
    both = np.load(tmp_path / "rank0.npz")
    k = list(both["idx"]).index(2)
    rel = np.linalg.norm(both["lat"][k] - solo[0]) / np.linalg.norm(solo[0])
    assert rel < 5e-3, rel


def test_product_entry_two_ranks_equal_single_process_bitwise(tmp_path):
    """The product-level 8-GPU mode on two ranks sharing one GPU: load_all_models_sharded (rank 0 alone reads the
    checkpoint; UNet + VAE in ONE broadcast) -> generate_batch_images_sharded (round-robin prompts, per-prompt seeds, uint8
    gather on rank 0).  The gathered images must equal, BITWISE, what this process computes for the same shards from its own
    load of the same checkpoint -- nothing depends on the rank, and the broadcast is lossless."""
    import stubs
    import dist_product_worker as W
    from layoutllm_t2i_amd import interface as itf
    from layoutllm_t2i_amd.arch import TINY, VAE_TINY
    from layoutllm_t2i_amd.dist import shard_indices
    ckpt = str(tmp_path / "tiny_gligen.pth")
    stubs.write_synthetic_checkpoint(ckpt, TINY, VAE_TINY, max_relations=10)
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_product_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, worker, ckpt, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-3000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["ckpt_reads"] >= 1 and b["ckpt_reads"] == 0, "only rank 0 may read the checkpoint"
    assert a["text_encoder"] and not b["text_encoder"] and a["n_images"] == len(W.PROMPTS) and b["n_images"] is None
    got = np.load(tmp_path / "images.npz")["imgs"]
    assert got.shape[0] == len(W.PROMPTS) and got.dtype == np.uint8 and got.shape[-1] == 3
    # the same shards in THIS process
    dev = "cuda:0"
    stubs.install_fake_sng_parser()
    am = itf.load_all_models(ckpt, dev)
    cond = itf.prepare_conditioning(am, W.PROMPTS, W.PHRASES, W.BOXES, stubs.toy_clip().to(dev), stubs.ToyProcessor(), dev)
    for r in range(world):
        mine = shard_indices(len(W.PROMPTS), r, world)
        ref = itf.run_shard(am, {k: v[mine] for k, v in cond.items()}, itf.prompt_noise([W.SEEDS[i] for i in mine], W.LATENT), dev,
                            steps=W.STEPS)
        assert np.array_equal(got[mine], ref), f"rank {r}: {np.abs(got[mine].astype(int) - ref.astype(int)).max()}"
    # single-process call of the same entry: one batch of all prompts; per-prompt seeds keep every image close to its sharded twin
    solo = itf.generate_batch_images_sharded(am, W.PROMPTS, W.PHRASES, W.BOXES, stubs.toy_clip().to(dev), stubs.ToyProcessor(), device=dev,
                                             seeds=W.SEEDS, steps=W.STEPS, latent=W.LATENT)
    solo = np.stack([np.asarray(im) for im in solo])
    assert solo.shape == got.shape and np.mean(np.abs(solo.astype(int) - got.astype(int)) <= 3) > 0.97


def test_product_entry_two_ranks_hip_text_encoder_each_rank_encodes_its_own_shard(tmp_path):
    """Same product entry with a checkpoint whose text encoder is a CLIP text tower: the tower's weights travel in the
    one bundle broadcast, rank 1 (which never opens the checkpoint) holds a HipCLIPTextEncoder, rank 0 only TOKENISES, and every
    rank encodes + denoises + decodes its own prompts.  Rank r's images must equal, bitwise, this process's own
    tokenize -> encode_conditioning(rows of rank r) -> run_shard from its own load of the checkpoint."""
    import stubs
    import dist_product_worker as W
    from layoutllm_t2i_amd import interface as itf
    from layoutllm_t2i_amd.arch import TINY, VAE_TINY
    from layoutllm_t2i_amd.dist import shard_indices
    ckpt = str(tmp_path / "tiny_gligen_clip.pth")
    stubs.write_synthetic_checkpoint(ckpt, TINY, VAE_TINY, max_relations=10, clip_text_tower=True)
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_product_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2",
                   W_CLIP_TOWER="1")
        procs.append(subprocess.Popen([sys.executable, worker, ckpt, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-3000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["ckpt_reads"] >= 1 and b["ckpt_reads"] == 0, "only rank 0 may read the checkpoint"
    assert a["text_encoder_type"] == b["text_encoder_type"] == "HipCLIPTextEncoder"
    got = np.load(tmp_path / "images.npz")["imgs"]
    assert got.shape[0] == len(W.PROMPTS) and got.dtype == np.uint8
    dev = "cuda:0"
    stubs.install_fake_sng_parser()
    am = itf.load_all_models(ckpt, dev)
    tok = itf.tokenize_conditioning(am, W.PROMPTS, W.PHRASES, W.BOXES, stubs.ToyProcessor())
    for r in range(world):
        mine = shard_indices(len(W.PROMPTS), r, world)
        cond = itf.encode_conditioning(am, tok, mine, am[2], dev)
        ref = itf.run_shard(am, cond, itf.prompt_noise([W.SEEDS[i] for i in mine], W.LATENT), dev, steps=W.STEPS)
        assert np.array_equal(got[mine], ref), f"rank {r}: {np.abs(got[mine].astype(int) - ref.astype(int)).max()}"


def test_product_entry_two_ranks_strict_mode(tmp_path):
    """load_all_models_sharded(..., strict=True) on rank 0 only: the SPLIT weight layout and the strict flag travel in the one bundle broadcast, rank 1
    (which never opens the checkpoint) runs its shard in strict mode too, and the gathered images equal -- bitwise -- this process's own strict load."""
    import stubs
    import dist_product_worker as W
    from layoutllm_t2i_amd import interface as itf
    from layoutllm_t2i_amd.arch import TINY, VAE_TINY
    from layoutllm_t2i_amd.dist import shard_indices
    ckpt = str(tmp_path / "tiny_gligen.pth")
    stubs.write_synthetic_checkpoint(ckpt, TINY, VAE_TINY, max_relations=10)
    world, port = 2, _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_product_worker.py")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2", W_STRICT="1")
        procs.append(subprocess.Popen([sys.executable, worker, ckpt, str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    try:
        for p in procs:
            out, err = p.communicate(timeout=600)
            assert p.returncode == 0, err[-3000:]
            res.append(json.loads(next(l for l in out.splitlines() if l.startswith("RESULT "))[7:]))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    a, b = sorted(res, key=lambda d: d["rank"])
    assert a["ckpt_reads"] >= 1 and b["ckpt_reads"] == 0
    assert a["strict"] and b["strict"] and a["split_weights"] and b["split_weights"], (a, b)
    got = np.load(tmp_path / "images.npz")["imgs"]
    dev = "cuda:0"
    stubs.install_fake_sng_parser()
    am = itf.load_all_models(ckpt, dev, strict=True)
    cond = itf.prepare_conditioning(am, W.PROMPTS, W.PHRASES, W.BOXES, stubs.toy_clip().to(dev), stubs.ToyProcessor(), dev)
    for r in range(world):
        mine = shard_indices(len(W.PROMPTS), r, world)
        ref = itf.run_shard(am, {k: v[mine] for k, v in cond.items()}, itf.prompt_noise([W.SEEDS[i] for i in mine], W.LATENT), dev, steps=W.STEPS)
        assert np.array_equal(got[mine], ref), f"rank {r}: {np.abs(got[mine].astype(int) - ref.astype(int)).max()}"
    # and the strict images differ from the default mode's (the flag really travelled)
    am0 = itf.load_all_models(ckpt, dev)
    ref0 = itf.run_shard(am0, cond, itf.prompt_noise(W.SEEDS, W.LATENT), dev, steps=W.STEPS)
    assert not np.array_equal(got, ref0)


def test_rccl_path_of_the_bench_at_world_1():
    """The multi-rank code path of bench.py (RCCL process group bound to the device, bundle broadcast of UNet + VAE, barriers,
    max-over-ranks all-reduce) launched exactly as the driver launches N > 1, with ONE rank: every collective goes through
    RCCL on the GPU, which the gloo tests cannot show.  The JSON line must come out and carry the broadcast time."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--tiny", "--steps", "1",
           "--warmup", "1", "--plms-steps", "4", "--no-cpu-baseline", "--no-hot-kernel"]
    p = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    line = next(l for l in p.stdout.splitlines() if l.startswith("{"))
    doc = json.loads(line)
    assert doc["n_gpus"] == 1 and doc["value"] > 0 and doc.get("weight_broadcast_ms") is not None, doc
