"""Worker of tests/test_bench_extras_cpu.py: N gloo ranks on CPU drive bench.run_extras() with extras that succeed, fail on one
rank before their collectives, fail without any collective, and are made to fail by BENCH_INJECT_FAILURE."""
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (module level: no GPU, no torch.cuda)


def main():
    dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    rank, world = dist.get_rank(), dist.get_world_size()

    def gather(obj):
        got = [None] * world
        dist.all_gather_object(got, obj)
        return got

    def summed(tag):
        def fn(w, r, ready):
            ready()
            t = torch.tensor([r + 1], dtype=torch.int64)
            dist.all_reduce(t)                      # the extra's own collective: reached by all ranks or by none
            return {"value": int(t.item()), "tag": tag}
        return fn

    def fails_before_its_collective(w, r, ready):
        if r == 2:
            raise MemoryError("rank 2 cannot allocate its shard")
        ready()
        t = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(t)
        return {"value": int(t.item())}

    def local_only(w, r, ready):                    # never calls ready(): the wrapper does
        if r == 1:
            raise ValueError("rank 1: local check failed")
        return {"value": r}

    extras = [("A_ok", summed("a")), ("B_fails_on_2", fails_before_its_collective), ("C_local_only", local_only), ("D_injected", summed("d")),
              ("E_ok_again", summed("e"))]
    res = bench.run_extras(extras, (world, rank), gather, rank)
    every = gather(res)
    if rank == 0:
        print("EXTRAS_JSON " + json.dumps(every))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
