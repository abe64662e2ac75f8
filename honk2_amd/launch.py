"""Start the N ranks of a multi-GPU run from the one command the user typed.

The reference selects its GPU count from ONE integer inside ONE command (``num_gpu`` in the config, ``run/test.py:69-70``,
``utils/torch_utils.py:9-22``: ``DataParallel`` threads).  Here a multi-GPU run is one process per GPU over RCCL, so the
command the user typed becomes a LAUNCHER: it starts ``torch.distributed.run`` as an ordinary child process (rendezvous on
127.0.0.1 at a free port), lets the ranks' stdout / stderr through, and returns the worst exit code.

This file imports nothing but the standard library and must stay that way: ``bench.py`` loads it by path before torch is
imported, and a launcher must never have made a HIP call (a process that has initialised the GPU may not be replaced or
forked into ranks on this platform -- the ranks are FRESH children).
"""
import os
import socket
import subprocess
import sys

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK")


def under_launcher():
    """True inside a rank that torchrun (or this launcher) started."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, script=None, argv=(), module=None, timeout_s=None, env=None):
    """Run ``n`` ranks of ``script`` (a path) or of ``-m module`` on this node; returns the exit code (0 only when every
    rank returned 0; 124 on timeout)."""
    if (script is None) == (module is None):
        raise ValueError("give a script path or a module name")
    if under_launcher():
        raise RuntimeError("launch_ranks called from inside a rank")
    env = dict(os.environ if env is None else env)
    for k in RANK_ENV:
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n)),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += ["-m", module] if module else [script]
    cmd += list(argv)
    try:
        return subprocess.run(cmd, env=env, timeout=timeout_s).returncode
    except subprocess.TimeoutExpired:
        print(f"launch: the {n}-rank run did not finish within {timeout_s} s", file=sys.stderr)
        return 124
