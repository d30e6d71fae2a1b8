"""Process environment of a multi-GPU (one process per GPU, RCCL over xGMI) run - ONE place for every launcher of ranks:
bench.py's self-spawn path, ranks started by an external torch.distributed.run, and the trainers' mp.spawn
(reference: direct_regression/train_direct_4gpu.py:311-339, progressive_cascade/train_progressive_4gpu.py main())."""
import os


def ensure_rccl_env(env=None):
    """HSA_ENABLE_IPC_MODE_LEGACY=0: the MI355X host driver of this pool only supports dmabuf IPC; without it RCCL (and any
    sharing of device tensors between processes) fails with `hipIpcGetMemHandle: invalid argument`.  Must be in the
    environment before a process makes its first HIP call - call this at the top of main(), and on the env dict handed to
    child processes.  An explicit user setting wins."""
    env = os.environ if env is None else env
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env
