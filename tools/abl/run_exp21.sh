#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_BWD=1 IA_ATTN_FWD=2
./attn_dev_d64.bin 48 385 12 1 0 1 0 6 | grep -v "rel err" | head -150
