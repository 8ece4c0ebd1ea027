#!/usr/bin/env python3
"""Recall@k / MRR@k of one ranked TSV against the ground truth -- same argv and output as
the reference's MEVI/evaluate.py:160-170 (drop-in for README step "evaluate")."""
from argparse import ArgumentParser

from mevi_amd.metrics import evaluate_main

if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("--dir_path", type=str, required=True)
    parser.add_argument("--gt_file", type=str, required=True)
    parser.add_argument("--ance_file", type=str, required=True)
    parser.add_argument("--recall_num", type=str, default="1,5,10,20,50,100,1000")
    parser.add_argument("--ofile", type=str, default=None)
    args = parser.parse_args()
    evaluate_main(args.dir_path, args.gt_file, args.ance_file, args.recall_num, args.ofile)
