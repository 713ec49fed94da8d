"""Can two ranks of an RCCL communicator share ONE GPU?  (1-GPU boxes are all the builder ever gets.)  Run: two processes, both on
cuda:0, one all_gather_into_tensor; prints what RCCL says.  Bounded by the caller's `timeout`."""
import datetime
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_DEBUG="WARN")
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60),
                                device_id=torch.device("cuda:0"))
        x = torch.full((4,), float(rank), device="cuda:0")
        out = torch.empty(4 * world, device="cuda:0")
        dist.all_gather_into_tensor(out, x)
        torch.cuda.synchronize()
        print(f"rank {rank}: all_gather ok {out.tolist()}", flush=True)
    except Exception as e:                                   # noqa: BLE001
        print(f"rank {rank}: FAILED {type(e).__name__}: {str(e)[:400]}", flush=True)
    finally:
        if dist.is_initialized():
            try:
                dist.destroy_process_group()
            except Exception:                                # noqa: BLE001
                pass


if __name__ == "__main__":
    mp.spawn(worker, args=(2, int(sys.argv[1]) if len(sys.argv) > 1 else 29531), nprocs=2, join=True)
