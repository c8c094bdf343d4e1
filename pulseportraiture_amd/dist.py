"""Sharding of subints over the GPUs of a node and the single gather of results.

Subints (and archives) are independent units (pptoas.py:247,344 are plain
loops), so ranks never exchange data during the fit; the only collective is one
gather of fixed-size result records to rank 0 (RCCL over xGMI when the backend
is "nccl"; "gloo" in the CPU tests)."""
import numpy as np

# columns of a result record
RECORD_FIELDS = (["phi", "DM", "GM", "tau", "alpha"] +
                 ["phi_err", "DM_err", "GM_err", "tau_err", "alpha_err"] +
                 ["nu_DM", "nu_GM", "nu_tau", "chi2", "red_chi2", "snr", "nfeval",
                  "return_code"])
RECORD_WIDTH = len(RECORD_FIELDS)


def shard_range(n, rank, world):
    """Contiguous block [lo, hi) of n units owned by `rank` (sizes differ by at
    most one; the first n % world ranks get the extra unit)."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_records(res):
    """dict of Engine.fit_batch arrays -> float64 [nsub, RECORD_WIDTH]."""
    n = res["params"].shape[0]
    rec = np.empty((n, RECORD_WIDTH))
    rec[:, 0:5] = res["params"]
    rec[:, 5:10] = res["param_errs"]
    rec[:, 10:13] = res["nu_refs"]
    rec[:, 13] = res["chi2"]
    rec[:, 14] = res["red_chi2"]
    rec[:, 15] = res["snr"]
    rec[:, 16] = res["nfeval"]
    rec[:, 17] = res["return_code"]
    return rec


def unpack_records(rec):
    return {name: rec[:, j] for j, name in enumerate(RECORD_FIELDS)}


def gather_records(rec, counts=None, device=None, dst=0):
    """Gather per-rank record blocks on rank `dst`; returns the concatenated
    [sum(counts), RECORD_WIDTH] array there and None elsewhere.  `rec` is a NumPy
    array or a torch tensor (a CUDA tensor is gathered as it is: device buffers
    over RCCL, no host staging; the result is then a tensor on the same device).
    Without an initialised process group this is the identity."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return rec
    world, rank = dist.get_world_size(), dist.get_rank()
    if counts is None:
        counts = [rec.shape[0]] * world
    nmax = max(counts)
    as_tensor = torch.is_tensor(rec)
    if as_tensor:
        # (gloo gathers host tensors only: a world of ranks sharing one GPU, in tests)
        dev = torch.device("cpu") if dist.get_backend() == "gloo" else rec.device
        src = rec.to(device=dev, dtype=torch.float64)
    else:
        dev = device if device is not None else "cpu"
        src = torch.from_numpy(np.ascontiguousarray(rec, dtype=np.float64)).to(dev)
    if src.shape[0] == nmax:
        buf = src.contiguous()
    else:
        buf = torch.zeros((nmax, RECORD_WIDTH), dtype=torch.float64, device=dev)
        buf[:src.shape[0]] = src
    outs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gather_list=outs, dst=dst)
    if rank != dst:
        return None
    cat = torch.cat([o[:c] for o, c in zip(outs, counts)])
    return cat if as_tensor else cat.cpu().numpy()


def records_checksum(rec):
    """Order-independent digest of a block of records (sum of every column and the
    number of rows): what a multi-GPU job prints so that runs can be compared."""
    a = rec.detach().cpu().numpy() if hasattr(rec, "detach") else np.asarray(rec)
    return {"rows": int(a.shape[0]), "column_sums": [float(v) for v in a.sum(axis=0)]}
