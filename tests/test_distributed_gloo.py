"""CPU suite: the N>1 path (volume sharding + barrier + max-over-ranks timing) rehearsed with world_size 2 on gloo."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import time
    import torch.distributed as dist
    from jointimagegeneration_amd import distributed as ggd
    r, w = ggd.init("gloo")
    assert (r, w) == (rank, world)
    mine = ggd.shard(7, r, w)
    gathered = [None] * w
    dist.all_gather_object(gathered, mine)
    # rank 1 "works" longer: the reported time must be the MAX over ranks on every rank
    elapsed = ggd.timed_region(lambda: time.sleep(0.05 + 0.25 * rank))
    q.put((rank, mine, gathered, elapsed))
    ggd.finalize()


def test_two_rank_sharding_and_timing():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (_, m0, g0, e0), (_, m1, g1, e1) = res
    assert m0 == [0, 2, 4, 6] and m1 == [1, 3, 5]                      # volume_id mod world_size
    assert g0 == g1 == [m0, m1]
    assert sorted(m0 + m1) == list(range(7))                           # disjoint cover: no volume sampled twice or dropped
    assert e0 >= 0.29 and e1 >= 0.29 and abs(e0 - e1) < 1e-6           # max over ranks, identical everywhere


def test_single_process_defaults():
    sys.path.insert(0, ROOT)
    from jointimagegeneration_amd import distributed as ggd
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    assert ggd.env_rank_world() == (0, 0, 1)
    assert ggd.shard(3, 0, 1) == [0, 1, 2]
    assert ggd.timed_region(lambda: None) >= 0.0


def test_nifti_writer_roundtrip(tmp_path):
    import gzip, struct
    import numpy as np
    sys.path.insert(0, ROOT)
    from jointimagegeneration_amd.io import write_nifti
    a = (np.arange(2 * 3 * 4) % 12).astype(np.uint8).reshape(2, 3, 4)
    p = str(tmp_path / "x.nii.gz")
    write_nifti(p, a)
    raw = gzip.open(p, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 348 and raw[344:347] == b"n+1"
    assert struct.unpack_from("<8h", raw, 40)[:4] == (3, 4, 3, 2)
    assert np.array_equal(np.frombuffer(raw[352:], dtype=np.uint8).reshape(2, 3, 4), a)


def _run_bench(args, env_extra, timeout=180):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GG_BENCH_T0")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_gpus2_spawns_its_own_ranks():
    """`bench.py --gpus 2` started WITHOUT a launcher becomes a parent that starts 2 ranks (rehearsed on CPU: gloo ranks, no
    sampling) and forwards rank 0's single JSON line; steps/warmup are reported as run, with the requested values beside them."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"GG_BENCH_DRY": "1", "GG_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["steps_requested"] == 3 and j["warmup"] == 1 and j["dry_run"] is True
    assert j["ms_per_step"] * j["steps"] <= j["wall_s_total"] * 1e3


def test_bench_budget_bounds_the_timed_volumes():
    import json
    r = _run_bench(["--gpus", "1", "--steps", "100000", "--warmup", "5", "--budget-s", "14"], {"GG_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert 1 <= j["steps"] < 100000 and j["steps_requested"] == 100000 and j["wall_s_total"] < 14.0


def test_bench_gpus_mismatch_and_failed_rank_are_errors():
    # a launcher that started the wrong number of ranks
    r = _run_bench(["--gpus", "4", "--steps", "1"], {"GG_BENCH_DRY": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
    # real (non-rehearsal) ranks on a box without a GPU fail, and the parent reports it
    if not torch.cuda.is_available():
        r = _run_bench(["--gpus", "2", "--steps", "1"], {"GG_DIST_BACKEND": "gloo"})
        assert r.returncode != 0
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_nifti_reader_is_the_inverse_of_the_writer(tmp_path):
    """io.read_nifti (the stage-1 -> stage-2 hand-off reads what ddpm_eval wrote; README.md:21, sample_diffusion.py:199): [D, H, W] arrays of
    every supported dtype through .nii and .nii.gz, and loud errors on what is not a little-endian single-file NIfTI-1 volume."""
    import numpy as np
    import pytest
    sys.path.insert(0, ROOT)
    from jointimagegeneration_amd.io import read_nifti, write_nifti
    rng = np.random.default_rng(3)
    for dt in (np.uint8, np.int16, np.int32, np.float32):
        a = (rng.random((5, 6, 7)) * 200).astype(dt)
        for ext in (".nii", ".nii.gz"):
            f = str(tmp_path / f"v_{np.dtype(dt).name}{ext}")
            write_nifti(f, a)
            b = read_nifti(f)
            assert b.dtype == a.dtype and b.shape == (5, 6, 7) and np.array_equal(a, b)
    bad = tmp_path / "bad.nii"
    bad.write_bytes(b"\0" * 400)
    with pytest.raises(ValueError, match="sizeof_hdr"):
        read_nifti(str(bad))
    f = str(tmp_path / "v_uint8.nii")
    raw = open(f, "rb").read()
    (tmp_path / "short.nii").write_bytes(raw[:-10])
    with pytest.raises(ValueError, match="truncated"):
        read_nifti(str(tmp_path / "short.nii"))


def test_pipeline_cli_shards_volumes_over_two_gloo_ranks(tmp_path):
    """`python -m jointimagegeneration_amd.pipeline --volumes 5` under two ranks (GG_PIPELINE_DRY=1: the sharding, seeds and file naming
    without a GPU): volume_id mod world_size, every volume exactly once, the seed of a volume independent of the world size."""
    import subprocess
    port = _free_port()
    out = tmp_path / "vols"
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GG_PIPELINE_DRY="1",
                   PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        procs.append(subprocess.Popen([sys.executable, "-m", "jointimagegeneration_amd.pipeline", "--volumes", "5", "--out", str(out)], env=env, cwd=ROOT))
    assert [p.wait(timeout=180) for p in procs] == [0, 0]
    got = {f: (out / f).read_text().split() for f in sorted(os.listdir(out))}
    assert sorted(got) == [f"ct_{v:04d}.txt" for v in range(5)]
    for v in range(5):
        words = got[f"ct_{v:04d}.txt"]                                 # "rank R world W seed S"
        assert int(words[1]) == v % 2 and int(words[3]) == 2 and int(words[5]) == 1024 + 1000 * v
    env1 = dict(os.environ, GG_PIPELINE_DRY="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env1.pop(k, None)
    out1 = tmp_path / "vols1"
    assert subprocess.call([sys.executable, "-m", "jointimagegeneration_amd.pipeline", "--volumes", "3", "--out", str(out1)], env=env1, cwd=ROOT) == 0
    assert [(out1 / f"ct_{v:04d}.txt").read_text().split()[5] for v in range(3)] == [str(1024 + 1000 * v) for v in range(3)]
