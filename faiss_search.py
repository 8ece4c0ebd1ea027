#!/usr/bin/env python3
"""Dense search CLI -- same argv, Python API and output file as the reference's
MEVI/faiss_search.py:80-98, served by the MI355X inner-product top-k instead of faiss-cpu.

`--param` is the faiss factory string: "Flat" is the exact search; "IVF<n>,Flat" (the default, as in the reference)
builds an IVF-Flat index with faiss's structure and defaults (mevi_amd/ivf.py; nprobe from MEVI_IVF_NPROBE, default 1)
and reports its recall against the exact search on stderr; any other index type is answered with EXACT search.
With torch.distributed initialised (torchrun) the corpus file is row-sharded across ranks and searched exactly.
"""
import argparse
import datetime
import os

import numpy as np

from mevi_amd.dense import is_trained_before_train, profile, search, shard_range, sharded_ip_topk  # noqa: F401  (API parity: search, profile)
from mevi_amd.io import map_rows, read, to_file  # noqa: F401
from mevi_amd.phases import finish, mark


def _distributed_search(query, doc_path, dim, topk):
    import torch
    import torch.distributed as dist

    from mevi_amd.dense import DenseIndex
    from mevi_amd.io import upload_rows

    rank, world = dist.get_rank(), dist.get_world_size()
    n_rows = os.path.getsize(doc_path) // (4 * dim)
    a, b = shard_range(n_rows, rank, world)
    shard = map_rows(doc_path, dim, rows=b - a, first_row=a)    # zero rows when there are fewer rows than ranks
    dev = torch.device("cuda", torch.cuda.current_device())
    index = DenseIndex(upload_rows(shard, dev))                  # this rank's rows of index.add(doc)
    q = torch.from_numpy(np.ascontiguousarray(query, dtype=np.float32)).to(dev)
    s, i = sharded_ip_topk(q, index, topk, id_offset=a)
    return s.cpu().numpy(), i.cpu().numpy()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--query_path", type=str, required=True)
    parser.add_argument("--doc_path", type=str, required=True)
    parser.add_argument("--output_path", type=str, required=True)
    parser.add_argument("--raw_query_path", type=str, required=True)
    parser.add_argument("--dim", type=int, default=768)
    parser.add_argument("--topk", type=int, default=1000)
    parser.add_argument("--param", type=str, default="IVF100,Flat")
    args = parser.parse_args()
    mark("start-up + imports")
    query = read(args.query_path, args.dim)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(os.environ.get("MEVI_DIST_BACKEND", "nccl"), timeout=datetime.timedelta(hours=24))
        if dist.get_rank() == 0:
            print(f"Param {args.param} trained: {is_trained_before_train(args.param)}.")
        dists, indices = _distributed_search(query, args.doc_path, args.dim, args.topk)
        if dist.get_rank() == 0:
            print(indices.dtype, indices.shape, dists.dtype, dists.shape)
            to_file(args.raw_query_path, args.output_path, dists, indices)
        dist.barrier()
        dist.destroy_process_group()
    else:
        doc = read(args.doc_path, args.dim)
        dists, indices = search(query, doc, args.dim, args.topk, args.param)
        mark("read + upload + index build + search")
        print(indices.dtype, indices.shape, dists.dtype, dists.shape)
        to_file(args.raw_query_path, args.output_path, dists, indices)
        mark("ranked TSV written")
        finish()
