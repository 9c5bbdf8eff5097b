"""The post-link check of videoswap_amd/build.py: the hand-scheduled kernels (persistent GEMM, flash attention, the 16-wave tile
kernel) count their own `vmcnt` entries, so a scratch reload inside them is one more VMEM operation in a queue the source has
budgeted by hand — the build refuses a library in which one of them spills or owns a private segment."""
import os

import pytest

from videoswap_amd import build

NOTE = """Displaying notes found in: .note
  Owner                Data size 	Description
  AMDGPU               0x00003c69	NT_AMDGPU_METADATA (AMDGPU Metadata)
    AMDGPU Metadata:
        ---
amdhsa.kernels:
  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           88
        .value_kind:     by_value
    .group_segment_fixed_size: 15360
    .name:           _ZN4vsxg12_GLOBAL__N_114gemm_pp_kernelILi2ELi2ELi1EEEvNS_10GemmParamsEi
    .private_segment_fixed_size: 16
    .sgpr_count:     106
    .sgpr_spill_count: 67
    .symbol:         _ZN4vsxg12_GLOBAL__N_114gemm_pp_kernelILi2ELi2ELi1EEEvNS_10GemmParamsEi.kd
    .vgpr_count:     256
    .vgpr_spill_count: 3
    .wavefront_size: 64
  - .agpr_count:     0
    .args:
      - .offset:         0
        .size:           136
        .value_kind:     by_value
      - .offset:         136
        .size:           4
        .value_kind:     hidden_block_count_x
    .group_segment_fixed_size: 0
    .name:           _ZN12_GLOBAL__N_115gn_apply_kernelEPKDF16_S1_liiiPKfS1_S1_iPDF16_
    .private_segment_fixed_size: 8
    .sgpr_count:     30
    .sgpr_spill_count: 0
    .symbol:         _ZN12_GLOBAL__N_115gn_apply_kernelEPKDF16_S1_liiiPKfS1_S1_iPDF16_.kd
    .vgpr_count:     40
    .vgpr_spill_count: 1
    .wavefront_size: 64
  - .agpr_count:     0
    .group_segment_fixed_size: 0
    .name:           _ZN12_GLOBAL__N_111gemm_kernelILi256ELi320ELi8ELi2ELb1ELi2EEEvN4vsxg10GemmParamsE
    .private_segment_fixed_size: 0
    .sgpr_spill_count: 46
    .vgpr_count:     124
    .vgpr_spill_count: 0
amdhsa.target:   amdgcn-amd-amdhsa--gfx950
...
"""


def test_notes_parser_and_offender_rule():
    res = build.parse_kernel_notes(NOTE)
    assert len(res) == 3
    pp = res['_ZN4vsxg12_GLOBAL__N_114gemm_pp_kernelILi2ELi2ELi1EEEvNS_10GemmParamsEi']
    assert pp == dict(agpr_count=0, group_segment_fixed_size=15360, private_segment_fixed_size=16, sgpr_count=106,
                      sgpr_spill_count=67, vgpr_count=256, vgpr_spill_count=3)
    bad = build.scratch_offenders(res)
    # the persistent kernel is refused; the GroupNorm kernel is not on the watch list; the 16-wave tile kernel is clean
    assert len(bad) == 1 and 'gemm_pp_kernel' in bad[0][0] and bad[0][1:] == (3, 16)
    # a private segment without spills (a dead stack object) counts as well
    pp['vgpr_spill_count'] = 0
    assert build.scratch_offenders(res)[0][1:] == (0, 16)
    pp['private_segment_fixed_size'] = 0
    assert build.scratch_offenders(res) == []


@pytest.mark.skipif(not os.path.exists(build.LIB), reason='libvsx.so not built')
def test_built_library_has_no_scratch_in_the_hand_scheduled_kernels():
    if not os.path.exists(os.path.join(build.LLVM_BIN, 'llvm-readelf')):
        pytest.skip('no llvm-readelf in this image')
    res = build.check_no_scratch()                  # raises with the offenders' names
    watched = [n for n in res if any(p in n for p in build.NO_SCRATCH)]
    assert len(watched) >= 40                       # every persistent kind + the flash-attention kernels + the 16-wave tiles
    assert all(res[n]['vgpr_spill_count'] == 0 and res[n]['private_segment_fixed_size'] == 0 for n in watched)
