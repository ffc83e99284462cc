#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python scratch/dbg_fp16_bisect.py tiny_concat fp16 1024 2>&1 | grep -v Warn | tail -16
timeout -k 10 200 python scratch/dbg_fp16_bisect.py tiny_concat bf16 1 2>&1 | grep -v Warn | grep "linear 3"
