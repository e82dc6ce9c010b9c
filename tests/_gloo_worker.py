"""Child process of tests/test_runner_cpu.py: one gloo rank that evaluates its shard and gathers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, n_items = (int(v) for v in sys.argv[1:5])
    mode = sys.argv[5] if len(sys.argv) > 5 else "pairs"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from keypoint_bench_amd import runner as rn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if mode == "vo":        # a sequence task: contiguous chunks, 13-wide rows, the pose chain composed from the gathered rows
        import numpy as np
        from vo_rows import vo_rows
        allrows = vo_rows(n_items)
        idx = rn.shard_chunk(n_items, rank, world)
        rows = rn.gather_rows(rn.pack_rows([allrows[i] for i in idx], n_items, rank, world), n_items, shard=rn.shard_chunk)
        agg = rn.aggregate("visual_odometer", rows)
        dist.barrier()
        dist.destroy_process_group()
        print("RESULT " + json.dumps({"rank": rank, "rows": rows[:, :13].tolist(), "t_end": agg["t_est"][-1].ravel().tolist(),
                                      "r_end": agg["r_est"][-1].ravel().tolist()}))
        return
    vals = [[i, 2 * i + 1, (i * 7) % 5] for i in rn.shard_indices(n_items, rank, world)]
    rows = rn.gather_rows(rn.pack_rows(vals, n_items, rank, world), n_items)
    agg = rn.aggregate("match_stats", rows)
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps({"rank": rank, "rows": rows[:, :3].tolist(), "agg": agg}))


if __name__ == "__main__":
    main()
