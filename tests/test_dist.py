"""CPU, world_size 2, gloo: the N > 1 path -- contiguous clip sharding + rank-ordered all-gather of logits +
counter reduction -- exercised with the same helpers bench.py and run/test.py use on RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from honk2_amd import dist_utils


def test_shard_bounds_cover_and_balance():
    for n in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            spans = [dist_utils.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert dist_utils.shard_bounds(65536, 3, 8) == (24576, 32768)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, w = dist_utils.init_from_env("gloo")
    assert (r, w) == (rank, world)
    from honk2_amd.metric import Acc, PerClassAcc
    from honk2_amd.run.test import reduce_results
    # every rank derives the same global "logits" and keeps its contiguous shard, like bench.py / run.test.main
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(total, 12, generator=g)
    target = torch.randint(0, 12, (total,), generator=g)
    lo, hi = dist_utils.shard_bounds(total, rank, world)
    counts = [b - a for a, b in (dist_utils.shard_bounds(total, i, world) for i in range(world))]
    gathered = dist_utils.all_gather_rows(logits[lo:hi].clone(), counts if len(set(counts)) > 1 else None)
    assert torch.equal(gathered, logits)                       # rank order == clip order
    # evaluation: rank r owns a contiguous run of the single-process loader's BATCHES (run/test.py:shard_loader); the
    # reduced loss must be the single-process mean of per-batch means for uneven splits, ragged last batches and
    # ranks that own no batch at all
    import torch.nn.functional as F
    bs = 8
    nb = (total + bs - 1) // bs
    lo_b, hi_b = dist_utils.shard_bounds(nb, rank, world)
    acc, pca = Acc(), PerClassAcc()
    loss_sum = 0.0
    for i in range(lo_b, hi_b):
        o, t = logits[i * bs:(i + 1) * bs], target[i * bs:(i + 1) * bs]
        loss_sum += F.cross_entropy(o, t).item()
        acc.accumulate(o, t)
        pca.accumulate(o, t)
    res = reduce_results(loss_sum, hi_b - lo_b, {"Acc": acc, "PerClassAcc": pca},
                         {i: f"c{i}" for i in range(12)}, torch.device("cpu"))
    ref_a, ref_p = Acc(), PerClassAcc()
    ref_a.accumulate(logits, target)
    ref_p.accumulate(logits, target)
    assert res["metric_Acc"] == ref_a.get_metric()
    assert res["metric_PerClassAcc"] == {f"c{k}": v for k, v in ref_p.get_metric().items()}
    ref_loss = sum(F.cross_entropy(logits[i * bs:(i + 1) * bs], target[i * bs:(i + 1) * bs]).item() for i in range(nb)) / nb
    assert abs(res["loss"] - ref_loss) < 1e-12
    np.save(os.path.join(out_dir, f"ok{rank}.npy"), np.ones(1))
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [64, 37, 5])
def test_two_rank_gather_and_reduce(tmp_path, total):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    assert os.path.exists(tmp_path / "ok0.npy") and os.path.exists(tmp_path / "ok1.npy")


def _write(path, text):
    with open(path, "w") as f:
        f.write(text)
    return str(path)


def test_one_command_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` / `python -m honk2_amd.run.test` with num_gpu > 1 become launchers (honk2_amd/launch.py): N fresh ranks
    under torch.distributed.run on 127.0.0.1, exit code = the worst rank's.  Reference: one command, one integer (run/test.py:69-70)."""
    from honk2_amd import launch
    ok = _write(tmp_path / "ok.py",
                "import os, sys\n"
                f"sys.path.insert(0, {ROOT!r})\n"
                "from honk2_amd import dist_utils\n"
                "import torch, torch.distributed as dist\n"
                "r, w = dist_utils.init_from_env('gloo')\n"
                "t = torch.tensor([r + 1.0]); dist.all_reduce(t)\n"
                "open(os.path.join(sys.argv[1], f'rank{r}.txt'), 'w').write(f'{w} {t.item()} {os.environ[\"MASTER_ADDR\"]}')\n"
                "dist.destroy_process_group()\n")
    assert launch.launch_ranks(2, script=ok, argv=[str(tmp_path)], timeout_s=300) == 0
    assert open(tmp_path / "rank0.txt").read() == "2 3.0 127.0.0.1" and open(tmp_path / "rank1.txt").read() == "2 3.0 127.0.0.1"
    bad = _write(tmp_path / "bad.py", "import os, sys\nsys.exit(3 if os.environ['RANK'] == '1' else 0)\n")
    assert launch.launch_ranks(2, script=bad, timeout_s=300) != 0                      # one failing rank fails the command
    with pytest.raises(ValueError):
        launch.launch_ranks(2)
    # bench.py's own front door: more than one GPU asked for and no rank environment -> it launches (no GPU here, so the ranks fail, loudly)
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in launch.RANK_ENV}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64", "--prewarm-ms", "0"],
                       env=dict(env, KWS_BENCH_BACKEND="gloo"), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "torch.distributed.run" not in r.stdout
    assert "launch with torch.distributed.run" not in r.stderr                            # the round-4 refusal is gone
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]                 # and no line is printed for a run that did not happen


def test_entry_point_rank_count_follows_num_gpu(monkeypatch):
    """ranks_for: the config's num_gpu clamped to the GPUs present (reference prepare_device, utils/torch_utils.py:9-22)."""
    from honk2_amd.run import test as entry
    monkeypatch.delenv("KWS_EVAL_RANKS", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert entry.ranks_for({"num_gpu": 3}) == 3 and entry.ranks_for({"num_gpu": 16}) == 8 and entry.ranks_for({"num_gpu": 0}) == 1
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    assert entry.ranks_for({"num_gpu": 4}) == 1
    monkeypatch.setenv("KWS_EVAL_RANKS", "2")
    assert entry.ranks_for({"num_gpu": 1}) == 2
    from honk2_amd import dist_utils
    monkeypatch.setenv("KWS_FORCE_DIST", "0")
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    assert not dist_utils.forced()                                                       # "0" is off
    monkeypatch.setenv("KWS_FORCE_DIST", "1")
    assert dist_utils.forced()
