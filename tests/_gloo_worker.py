"""Child process of tests/test_runner_cpu.py: one gloo rank that evaluates its shard and gathers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, n_items = (int(v) for v in sys.argv[1:5])
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from keypoint_bench_amd import runner as rn
    dist.init_process_group("gloo", rank=rank, world_size=world)
    vals = [[i, 2 * i + 1, (i * 7) % 5] for i in rn.shard_indices(n_items, rank, world)]
    rows = rn.gather_rows(rn.pack_rows(vals, n_items, rank, world), n_items)
    agg = rn.aggregate("match_stats", rows)
    dist.barrier()
    dist.destroy_process_group()
    print("RESULT " + json.dumps({"rank": rank, "rows": rows[:, :3].tolist(), "agg": agg}))


if __name__ == "__main__":
    main()
