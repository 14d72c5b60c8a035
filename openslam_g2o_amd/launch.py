"""Self-launch of the one-process-per-GPU benches: `python bench.py --gpus N` without a launcher in front of it
re-executes itself under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  When the driver (or a
user) already started the ranks -- WORLD_SIZE is set -- nothing happens here."""
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def relaunch_if_needed(gpus, script, argv=None):
    """Returns normally inside a rank (or for gpus == 1); otherwise spawns `gpus` ranks of `script` with the same
    arguments, forwards their output and exits with their return code."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(script)] + argv
    sys.stderr.write("launch: %d ranks via torch.distributed.run\n" % gpus)
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))
