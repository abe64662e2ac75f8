"""Evaluator entry point: ``python -m honk2_amd.run.test --config <honk2 json>`` (reference ``run/test.py``).

``evaluate`` keeps the reference signature and result format (``run/test.py:18-41``):
``{"loss": mean of per-batch mean CE, "metric_Acc": float, "metric_PerClassAcc": {class name: acc}}``.
When the model is a honk2_amd model on a GPU, the loss is ``ce_loss`` and the metrics are ``Acc`` /
``PerClassAcc``, the per-batch tail (reference ``:28-33``: ``loss.item()``, ``argmax``, ``.tolist()`` -- two to
three device syncs per batch) is replaced by the fused ``kws_eval_batch`` kernel and ONE device-to-host copy at
the end; any other loss/metric falls back to calling them per batch exactly as the reference does.

``main`` repairs what cannot run in the reference as written (SURVEY.md finding F10): metrics are instantiated,
``label_mapping`` is passed to ``evaluate``, ``evaluate_model_dir`` / ``best_dev_metric`` are optional.  Under
``torchrun`` (WORLD_SIZE > 1) every rank evaluates a contiguous shard and logits are all-gathered over RCCL.
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
        results = {"loss": total_loss / len(data_loader)}
        results.update(collect_metrics(metrics, label_mapping))
        return results

    inner = model.module if hasattr(model, "module") else model
    engine = inner.engine()
    n_labels = inner.config["n_labels"]
    n_batches = len(data_loader)
    stats = torch.zeros(2 + 2 * n_labels, dtype=torch.int64, device=device)
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
    per_batch = [s / b for s, b in zip(host_loss, sizes)]
    results = {"loss": sum(per_batch) / len(data_loader)}
    for metric in metrics.values():
        if isinstance(metric, Acc):
            metric.add_counts(host_stats[0], host_stats[1])
        else:
            metric.add_counts(host_stats[2:2 + n_labels], host_stats[2 + n_labels:2 + 2 * n_labels])
    results.update(collect_metrics(metrics, label_mapping))
    return results


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
    if device.type == "cuda" and world > 1:
        device = torch.device("cuda", torch.cuda.current_device())
    model = build_model(config).to(device)
    if rank == 0:
        print("model:", type(model).__name__, f"({model.num_params()} parameters)")

    test_data_loader = init_data_loader(config, DatasetType.TEST)
    label_mapping = test_data_loader.dataset.label_mapping
    if world > 1:   # one contiguous shard per rank; weights are replicated by construction (same seed / checkpoint)
        lo, hi = dist_utils.shard_bounds(len(test_data_loader.dataset), rank, world)
        shard = torch.utils.data.Subset(test_data_loader.dataset, range(lo, hi))
        shard.label_mapping = label_mapping
        test_data_loader = type(test_data_loader)(
            {"audio_preprocessing": test_data_loader.audio_preprocessing, "batch_size": test_data_loader.batch_size,
             "num_workers": test_data_loader.num_workers}, shard)
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

    results = evaluate(device, "test", model, test_data_loader, loss_fn, metrics, label_mapping, progress=rank == 0)
    if world > 1:
        results = reduce_results(results, metrics, label_mapping, device)
    if rank == 0:
        print("Test results")
        pprint(results)
    return results


def reduce_results(results, metrics, label_mapping, device):
    """Combine per-rank counters (sum) and losses (mean over ranks) with one small all-reduce."""
    import torch.distributed as dist
    n = len(label_mapping)
    buf = torch.zeros(3 + 2 * n, dtype=torch.float64, device=device)
    buf[0] = results["loss"]
    for m in metrics.values():
        if isinstance(m, Acc):
            buf[1], buf[2] = m.correct, m.total
        elif isinstance(m, PerClassAcc):
            for k, v in m.total.items():
                buf[3 + k] = m.correct[k]
                buf[3 + n + k] = v
    dist.all_reduce(buf)
    host = buf.cpu().tolist()
    for m in metrics.values():
        m.reset_metric()
        if isinstance(m, Acc):
            m.add_counts(round(host[1]), round(host[2]))
        elif isinstance(m, PerClassAcc):
            m.add_counts([round(x) for x in host[3:3 + n]], [round(x) for x in host[3 + n:3 + 2 * n]])
    out = {"loss": host[0] / dist.get_world_size()}
    out.update(collect_metrics(metrics, label_mapping))
    return out


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="MI355X-native honk2 evaluator")
    parser.add_argument("--config", default=None, required=True, type=str, help="path to a honk2 config file")
    main(load_json(parser.parse_args().config))
