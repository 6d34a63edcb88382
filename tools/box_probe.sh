#!/bin/bash
# Runs ON the GPU box: which kind of box is this (Env.step access shape with fresh allocations), and does the 16 GiB rule hold in ONE allocation?
./tools/membench 4194304 2>&1 | grep -E "Env.step shape, rows x4|float4 copy"
./tools/membench --bigsweep 4194304 2>&1 | awk 'NR%6==1' | cut -c1-60
./tools/membench --ballast 4194304 2>&1 | cut -c40-110
