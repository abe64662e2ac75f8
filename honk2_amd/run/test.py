"""Evaluator entry point: ``python -m honk2_amd.run.test --config <honk2 json>`` (reference ``run/test.py``).

``evaluate`` keeps the reference signature and result format (``run/test.py:18-41``):
``{"loss": mean of per-batch mean CE, "metric_Acc": float, "metric_PerClassAcc": {class name: acc}}``.
When the model is a honk2_amd model on a GPU, the loss is ``ce_loss`` and the metrics are ``Acc`` /
``PerClassAcc``, the per-batch tail (reference ``:28-33``: ``loss.item()``, ``argmax``, ``.tolist()`` -- two to
three device syncs per batch) is replaced by the fused ``kws_eval_batch`` kernel and ONE device-to-host copy at
the end; any other loss/metric falls back to calling them per batch exactly as the reference does.

``main`` repairs what cannot run in the reference as written (SURVEY.md finding F10): metrics are instantiated,
``label_mapping`` is passed to ``evaluate``, ``evaluate_model_dir`` / ``best_dev_metric`` are optional.  With ``num_gpu`` > 1
in the config (and that many GPUs present) the command starts one process per GPU itself (``cli`` -> ``launch.launch_ranks``; an
explicit ``torchrun`` works too): every rank evaluates a contiguous shard of the batches and the counters are all-reduced over RCCL.
"""
import argparse
import os
from pprint import pprint

import torch
from tqdm import tqdm

from .. import dist_utils
from ..dataset import DatasetType
from ..loss_function import ce_loss
from ..metric import Acc, PerClassAcc, collect_metrics
from ..model import BaseModel
from ..utils import find_cls, load_checkpoint_state, load_json, prepare_device
from .run_utils import init_data_loader, set_seed


def _fusable(device, model, loss_fn, metrics):
    inner = model.module if hasattr(model, "module") else model
    return (isinstance(inner, BaseModel) and torch.device(device).type == "cuda" and loss_fn is ce_loss
            and all(type(m) in (Acc, PerClassAcc) for m in metrics.values()))


def evaluate(device, prefix, model, data_loader, loss_fn, metrics, label_mapping, progress=True):
    """Reference signature and result dict (``run/test.py:18-41``)."""
    loss_sum, n_batches = evaluate_partial(device, prefix, model, data_loader, loss_fn, metrics, progress)
    results = {"loss": loss_sum / n_batches}               # an empty loader divides by zero, as in the reference
    results.update(collect_metrics(metrics, label_mapping))
    return results


def evaluate_partial(device, prefix, model, data_loader, loss_fn, metrics, progress=True):
    """One pass over `data_loader`: accumulates into `metrics` and returns (sum of the per-batch mean losses, number of
    batches) -- the two numbers a multi-rank evaluation has to add up before dividing once."""
    model.eval()
    batches = tqdm(data_loader, desc=f"Evaluating {prefix} dataset") if progress else data_loader
    if not _fusable(device, model, loss_fn, metrics):
        total_loss = 0
        for data, target in batches:
            data, target = data.to(device), target.to(device)
            with torch.no_grad():
                output = model(data)
            total_loss += loss_fn(output, target).item()
            for metric in metrics.values():
                metric.accumulate(output, target)
        return total_loss, len(data_loader)

    inner = model.module if hasattr(model, "module") else model
    engine = inner.engine()
    n_labels = inner.config["n_labels"]
    n_batches = len(data_loader)
    stats = torch.zeros(3 + 2 * n_labels, dtype=torch.int64, device=device)
    loss_sums = torch.zeros(max(n_batches, 1), dtype=torch.float64, device=device)
    sizes = []
    for i, (data, target) in enumerate(batches):
        data = data.to(device, non_blocking=True)
        target = target.to(device=device, dtype=torch.int64, non_blocking=True)
        output = inner.forward_wav(data) if data.dim() == 2 else inner(data)
        engine.eval_batch(output, target, stats, loss_sums[i:i + 1])
        sizes.append(output.shape[0])
    host_stats = stats.cpu().tolist()                      # the one synchronising copy
    host_loss = loss_sums.cpu().tolist()
    if host_stats[2 + 2 * n_labels]:                       # the reference's F.cross_entropy raises on such a target
        raise IndexError(f"{host_stats[2 + 2 * n_labels]} target(s) outside [0, {n_labels}) in the {prefix} dataset")
    per_batch = [s / b for s, b in zip(host_loss, sizes)]
    for metric in metrics.values():
        if isinstance(metric, Acc):
            metric.add_counts(host_stats[0], host_stats[1])
        else:
            metric.add_counts(host_stats[2:2 + n_labels], host_stats[2 + n_labels:2 + 2 * n_labels])
    return sum(per_batch), len(data_loader)


def build_model(config):
    n_labels = len(config["target_class"]) + int(bool(config["unknown_class"])) + int(bool(config["silence_class"]))
    model_config = config["model"]
    model_class = find_cls(f"model.{model_config['name']}")
    model_config["config"]["n_labels"] = n_labels
    return model_class(model_config["config"])


def main(config):
    rank, world = dist_utils.init_from_env()
    set_seed(config["seed"])
    device, _ = prepare_device(max(config["num_gpu"], 1))
    sharded = dist_utils.active()                      # world > 1, or a forced one-rank process group (dist_utils.forced)
    if device.type == "cuda" and sharded:
        device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(config).to(device)
    if rank == 0:
        print("model:", type(model).__name__, f"({model.num_params()} parameters)")

    test_data_loader = init_data_loader(config, DatasetType.TEST)
    label_mapping = test_data_loader.dataset.label_mapping
    if sharded:
        test_data_loader = shard_loader(test_data_loader, rank, world)
    if rank == 0:
        print(f"test dataset size: {len(test_data_loader.dataset)}" + (f" (per rank, {world} ranks)" if world > 1 else ""))

    loss_fn = find_cls(f"loss_fn.{config['loss_fn']}")
    metrics = {name: find_cls(f"metric.{name}")() for name in config["metric"]}

    model_dir = config.get("evaluate_model_dir")
    if model_dir:
        name = f"checkpoint_{config['evaluate_epoch']}.pt" if "evaluate_epoch" in config else "best_model.pt"
        state, extra = load_checkpoint_state(os.path.join(model_dir, name))
        model.load_state_dict(state)
        if rank == 0:
            print("Training results")
            for key in ("best_epoch", "best_dev_loss", "best_dev_criterion"):
                if key in extra:
                    print(f"\t{key}: {extra[key]}")
    elif rank == 0:
        print("no evaluate_model_dir in config: evaluating randomly initialised weights")

    if sharded:
        loss_sum, n_batches = evaluate_partial(device, "test", model, test_data_loader, loss_fn, metrics, progress=rank == 0)
        results = reduce_results(loss_sum, n_batches, metrics, label_mapping, device)
    else:
        results = evaluate(device, "test", model, test_data_loader, loss_fn, metrics, label_mapping, progress=True)
    if rank == 0:
        print("Test results")
        pprint(results)
    return results


def shard_loader(loader, rank, world):
    """Rank `rank`'s loader: a contiguous run of the single-process loader's BATCHES (so every batch, and with it every
    per-batch mean loss, is the one the single-process evaluation sees), built from the same loader config.  A rank may
    end up with no batch at all when the dataset has fewer batches than ranks."""
    if getattr(loader, "_shuffled", False):
        raise ValueError("sharded evaluation needs an unshuffled test loader")
    n, bs = len(loader.dataset), loader.batch_size
    lo_b, hi_b = dist_utils.shard_bounds((n + bs - 1) // bs, rank, world)
    shard = torch.utils.data.Subset(loader.dataset, range(min(lo_b * bs, n), min(hi_b * bs, n)))
    shard.label_mapping = loader.dataset.label_mapping
    return type(loader)(dict(loader.config, shuffle=False), shard)


def reduce_results(loss_sum, n_batches, metrics, label_mapping, device):
    """One small all-reduce (sum) of [sum of per-batch mean losses, batches, counters]; the loss is divided ONCE by the
    global number of batches, which reproduces the single-process ``total_loss / len(data_loader)`` for any split."""
    import torch.distributed as dist
    n = len(label_mapping)
    buf = torch.zeros(4 + 2 * n, dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    buf[0], buf[1] = loss_sum, n_batches
    for m in metrics.values():
        if isinstance(m, Acc):
            buf[2], buf[3] = m.correct, m.total
        elif isinstance(m, PerClassAcc):
            for k, v in m.total.items():
                buf[4 + k] = m.correct[k]
                buf[4 + n + k] = v
    dist.all_reduce(buf)
    host = buf.cpu().tolist()
    for m in metrics.values():
        m.reset_metric()
        if isinstance(m, Acc):
            m.add_counts(round(host[2]), round(host[3]))
        elif isinstance(m, PerClassAcc):
            m.add_counts([round(x) for x in host[4:4 + n]], [round(x) for x in host[4 + n:4 + 2 * n]])
    if round(host[1]) == 0:
        raise ZeroDivisionError("empty test dataset")
    out = {"loss": host[0] / round(host[1])}
    out.update(collect_metrics(metrics, label_mapping))
    return out


def ranks_for(config):
    """How many processes `python -m honk2_amd.run.test` runs as: the config's ``num_gpu`` clamped to the GPUs present, exactly the
    count the reference's ``prepare_device`` hands to ``DataParallel`` (``utils/torch_utils.py:9-22``, ``run/test.py:69-70``).
    (``torch.cuda.device_count()`` does not initialise the GPU on ROCm.)"""
    rehearse = os.environ.get("KWS_EVAL_RANKS")       # tests: N ranks on a box with fewer GPUs (with KWS_BENCH_BACKEND=gloo KWS_BENCH_ONE_DEVICE=1)
    if rehearse:
        return max(1, int(rehearse))
    return max(1, min(int(config.get("num_gpu", 1)), torch.cuda.device_count()))


def cli(argv=None):
    parser = argparse.ArgumentParser(description="MI355X-native honk2 evaluator")
    parser.add_argument("--config", default=None, required=True, type=str, help="path to a honk2 config file")
    args = parser.parse_args(argv)
    config = load_json(args.config)
    from .. import launch
    n = ranks_for(config)
    if n > 1 and not launch.under_launcher():
        # one process per GPU: this process has made no HIP call yet and becomes the launcher of n fresh ranks of this very command
        return launch.launch_ranks(n, module="honk2_amd.run.test", argv=["--config", args.config])
    main(config)
    return 0


if __name__ == "__main__":
    import sys
    sys.exit(cli())
