#!/usr/bin/env python3
"""Writes a copy of dsim_api.hip whose bucket-form neighbour query stamps its workgroups' phases (dev helper; the product
library carries none of this): with dsim_downwash_args.pairs_evaluated given, workgroup g of k_dw_query_cell writes
wall_clock64() at [8 + 6 g + k]: k = 0 start, 1 counts in (first barrier), 2 tile banded (set-up done), 3 / 4 wave 0 / 1 done,
5 = HW_ID.  The pair counting is switched off in this copy.  usage: make_stamped_query.py <out.hip>; build it like the
library (hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared) and run tools/c5_query_timeline.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
s = open(os.path.join(ROOT, "dronesim_amd", "csrc", "dsim_api.hip")).read()


def sub(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


sub('''  const unsigned t = threadIdx.x;
  {
    const long long gid = (long long)blockIdx.x * TPB + t;''', '''  const unsigned t = threadIdx.x;
  unsigned long long* const stamp = a.pairs ? a.pairs + 8 + 6ull * blockIdx.x : nullptr;
  if (stamp && t == 0) { stamp[0] = wall_clock64(); unsigned hw; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw)); stamp[5] = hw; }
  {
    const long long gid = (long long)blockIdx.x * TPB + t;''')
sub('''  __syncthreads();
  const int cnt_c = min(rcount, DW_CAP);''', '''  __syncthreads();
  if (stamp && t == 0) stamp[1] = wall_clock64();
  const int cnt_c = min(rcount, DW_CAP);''')
sub('''      __syncthreads();
      // ---- the groups, dealt to the waves in snake order ----''', '''      __syncthreads();
      if (stamp && t == 0) stamp[2] = wall_clock64();
      // ---- the groups, dealt to the waves in snake order ----''')
sub('''        if (have && sub8 == 0) dw_write(a, (long long)__float_as_int(me.w) - a.local_offset, K * fz, accumulate);
      }
      return;''', '''        if (have && sub8 == 0) dw_write(a, (long long)__float_as_int(me.w) - a.local_offset, K * fz, accumulate);
      }
      if (stamp && (t & 63u) == 0) stamp[3 + (t >> 6)] = wall_clock64();
      return;''')
sub("if (a.pairs && sub == 0) atomicAdd(a.pairs, (unsigned long long)cnt);", ";")
sub("if (a.pairs && sub == 0) atomicAdd(a.pairs, (unsigned long long)n_ovf_c);", ";")
sub("if (a.pairs) {                 // (wave-uniform)", "if (false) {                 // (wave-uniform)")
sub("if (a.pairs && have && sub_p == 0) atomicAdd", "if (false) atomicAdd")
open(sys.argv[1], "w").write(s)
