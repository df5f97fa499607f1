#!/usr/bin/env python3
"""Generates regprobe.s: a gfx950 kernel that fills SGPRs s8..s95 and VGPRs v2..v63 of every wave with known patterns,
waits (sleeps, loads, float atomics -- what hypercol_scatter_kernel does), then dumps EVERY register to memory.
Host side: tools/experiments/regprobe_host.hip.  Used for the shared-GPU finding of DESIGN.md 6: does a co-resident
process's bf16x3 GEMM change another process's live registers, and which ones?"""
NS, NV = 96, 64
L = []
A = L.append
A('\t.text\n\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"\n\t.globl regprobe\n\t.p2align 8\n\t.type regprobe,@function\nregprobe:')
A('\ts_load_dwordx2 s[4:5], s[0:1], 0x0\n\ts_load_dword s6, s[0:1], 0x8\n\ts_load_dwordx2 s[96:97], s[0:1], 0x10')
A('\tv_lshrrev_b32_e32 v1, 6, v0\n\ts_waitcnt lgkmcnt(0)\n\tv_readfirstlane_b32 s7, v1\n\ts_lshl_b32 s3, s2, 2\n\ts_add_u32 s7, s7, s3')
A(f'\ts_mul_i32 s3, s7, {(NS + NV) * 256}\n\ts_add_u32 s4, s4, s3\n\ts_addc_u32 s5, s5, 0')
A('\tv_and_b32_e32 v0, 63, v0\n\tv_lshlrev_b32_e32 v0, 2, v0')
for n in range(8, NS):
    A(f'\ts_mov_b32 s{n}, 0x{0x5A000000 | n:08x}')
for n in range(2, NV):
    A(f'\tv_mov_b32_e32 v{n}, 0x{0x7B000000 | n:08x}')
A('\tv_mov_b32_e32 v1, 1.0')
A('.Lspin:\n\ts_sleep 8\n\tglobal_atomic_add_f32 v0, v1, s[96:97]\n\tglobal_load_dword v1, v0, s[96:97] offset:1024\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v1, 1.0'
  '\n\ts_sub_u32 s6, s6, 1\n\ts_cmp_lg_u32 s6, 0\n\ts_cbranch_scc1 .Lspin')
for n in range(0, NV):
    A(f'\tglobal_store_dword v0, v{n}, s[4:5]\n\ts_add_u32 s4, s4, 256\n\ts_addc_u32 s5, s5, 0')
A('\ts_waitcnt vmcnt(0)')
for n in range(0, NS):
    A(f'\tv_mov_b32_e32 v2, s{n}\n\tglobal_store_dword v0, v2, s[4:5]\n\ts_add_u32 s4, s4, 256\n\ts_addc_u32 s5, s5, 0')
A('\ts_waitcnt vmcnt(0)\n\ts_endpgm')
A('.Lend:\n\t.size regprobe, .Lend-regprobe')
A('''\t.rodata
\t.p2align 6
\t.amdhsa_kernel regprobe
\t\t.amdhsa_group_segment_fixed_size 0
\t\t.amdhsa_private_segment_fixed_size 0
\t\t.amdhsa_kernarg_size 24
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_system_sgpr_workgroup_id_x 1
\t\t.amdhsa_system_vgpr_workitem_id 0
\t\t.amdhsa_next_free_vgpr 64
\t\t.amdhsa_next_free_sgpr 98
\t\t.amdhsa_accum_offset 64
\t\t.amdhsa_reserve_vcc 1
\t\t.amdhsa_float_denorm_mode_32 3
\t\t.amdhsa_float_denorm_mode_16_64 3
\t\t.amdhsa_dx10_clamp 1
\t\t.amdhsa_ieee_mode 1
\t.end_amdhsa_kernel
\t.amdgpu_metadata
---
amdhsa.kernels:
  - .args:
      - .address_space: global
        .offset: 0
        .size: 8
        .value_kind: global_buffer
      - .offset: 8
        .size: 4
        .value_kind: by_value
      - .offset: 12
        .size: 4
        .value_kind: by_value
      - .address_space: global
        .offset: 16
        .size: 8
        .value_kind: global_buffer
    .group_segment_fixed_size: 0
    .kernarg_segment_align: 8
    .kernarg_segment_size: 24
    .max_flat_workgroup_size: 256
    .name: regprobe
    .private_segment_fixed_size: 0
    .sgpr_count: 104
    .symbol: regprobe.kd
    .vgpr_count: 64
    .wavefront_size: 64
amdhsa.target: amdgcn-amd-amdhsa--gfx950
amdhsa.version:
  - 1
  - 2
...
\t.end_amdgpu_metadata''')
open(__import__('sys').argv[1], 'w').write('\n'.join(L) + '\n')
