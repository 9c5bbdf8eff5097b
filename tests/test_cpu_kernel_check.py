"""Kernels executed FROM THEIR SOURCE on the CPU (tools/cpu_check): the backward kernels of the training step and the
persistent MFMA GEMM / conv kernel with its schedule / option variants.

The backward kernels of the training step (videoswap_amd/csrc/train.hip) executed FROM THEIR SOURCE on the
CPU: tools/cpu_check/hip/hip_runtime.h maps the HIP execution model of these simple kernels (a workgroup = OS threads,
__syncthreads = a barrier, __shfl_xor = an exchange between two wave barriers) onto the host, tools/cpu_check/check_train.cpp
calls every C-ABI entry point and compares with a double-precision restatement of the formula.  This checks index
arithmetic and reduction logic without a GPU; the GPU tests (tests/test_autograd.py) remain the
parity tests proper."""
import os
import shutil
import subprocess

import pytest

from util import ROOT

CXX = os.environ.get('CXX_HOST', '/opt/rocm/lib/llvm/bin/clang++')


@pytest.mark.skipif(not (os.path.isfile(CXX) or shutil.which('clang++')), reason='needs clang++ (C++20, _Float16)')
def test_backward_kernels_run_from_source_on_the_cpu(tmp_path):
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    exe = str(tmp_path / 'check_train')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-o', exe,
           os.path.join(src, 'check_train.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout + r.stderr
    assert r.stdout.count(' ok') == 11


@pytest.mark.skipif(not (os.path.isfile(CXX) or shutil.which('clang++')), reason='needs clang++ (C++20, _Float16)')
def test_persistent_gemm_schedules_run_from_source_on_the_cpu(tmp_path):
    """videoswap_amd/csrc/gemm_pp.hip (the shipped persistent kernel) under tools/cpu_check/hip_gemm.h (MFMA as a wave
    collective with the hardware's fragment layout, LDS-DMA as a synchronous 16-byte-per-lane copy with the descriptor's
    range check, wave lockstep at the scheduling barriers): every epilogue kind the kernel is compiled for, under the
    default 2-D tile walk and under the linear one (pp_sched 8), must reproduce a double-precision GEMM / convolution, the
    two walks bit for bit alike, without a single read past a tensor.  (What this cannot see is the
    asynchronous ordering of the real DMA; the late-landing run below covers the other extreme of it.)  A subset of
    `make -C tools/cpu_check run`: one case per epilogue / loader kind."""
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    exe = str(tmp_path / 'check_gemm_pp')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-Wno-unused-value', '-o', exe,
           os.path.join(src, 'check_gemm_pp.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    runs = [('0', '0,8'),                        # one 256x320 tile, one slab
            ('4', '0'),                          # GEGLU epilogue
            ('8', '0'),                          # LayerNorm folded into the GEMM (rowscale / colvec), ragged M, residual ring
            ('9', '0'),                          # ... through the GEGLU epilogue (128-row)
            ('15', '0'),                         # 3x3 convolution, two sources, row-vector ring
            ('19', '0'),                         # 48 rows per vector: 32-row blocks that meet two row vectors
            ('21', '0'),                         # convolution without an addend, 128-row tiles
            ('22', '0,16,32'),                   # image rows of 32 pixels: ONE A slab per filter row read at three row offsets
                                                 # (default) = a private slab per tap (16), bit for bit; 32 = every CU walks the
                                                 # pieces of a slab in the same order
            ('18', '0,16'),                      # image rows of 16 pixels (round 5): left / right padding lanes INSIDE a 32-row block
            ('34', '0'),                         # sub-pixel form of the nearest-2x convolution: four classes of output pixels
            ('100', '0'), ('101', '0')]        # transposed store (V^T) of the persistent kernel, plain and with the LayerNorm identity
    if os.environ.get('VSX_CPU_CHECK_FULL'):     # a minute or more each: two sources, image rows as long as the tile, W = 24
        runs += [('7', '0,8'),                   # 36 tiles, tiles_n = 12: the 2-D walk against the linear one
                 ('23', '0,16,32'), ('31', '0,16'), ('33', '0,8'), ('35', '0')]
    # (`make -C tools/cpu_check run` walks every case: the remaining kernel kinds, stride 2, nearest-2x, K tails)
    for case, scheds in runs:
        r = subprocess.run([exe, case, scheds], capture_output=True, text=True, timeout=900)
        print(r.stdout)
        assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # ordering hazards: the multi-tile, multi-slab case again with every DMA piece landing as LATE as the kernel's own
    # waits allow (the default run above lands them at issue, the other extreme); see tools/cpu_check/hip_gemm.h
    for case, scheds in (('1', '0'), ('22', '0'), ('15', '0')):      # ('22': the shared A slab's own ring parity; '15': image rows of 8 pixels)
        r = subprocess.run([exe, case, scheds], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, CPUHIP_DMA='late'))
        print(r.stdout)
        assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_gemm_entry_point_runs_on_the_cpu(tmp_path):
    """vsx_gemm_f16 itself — descriptor checks, dispatch, the workgroup-per-tile kernels of csrc/gemm.hip (row-per-lane epilogue
    with the permlane32 exchange, prefetched residual and bias quads, GEGLU, the transposed V^T store), split-K with its combine
    kernel, the LayerNorm fold in every one of them, batched and convolution loaders — compiled from the real sources for the
    host and compared with double-precision references (tools/cpu_check/check_gemm_api.cpp).  VSX_TUNE_TILE forces the
    128x320 / 128x160 / 256x320 tile kernels, which problems this small would never reach through the dispatch."""
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    exe = str(tmp_path / 'check_gemm_api')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-Wno-unused-value', '-o', exe,
           os.path.join(src, 'check_gemm_api.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    env = {k: v for k, v in os.environ.items() if k not in ('VSX_TUNE_TILE', 'VSX_GEMM_PP', 'VSX_PP_SCHED', 'VSX_GEMM_WS')}
    env['CPUHIP_QUICK'] = '1'          # cases 21 / 22: a subset of their sub-cases (`make -C tools/cpu_check run` walks all)
    # (13: the persistent kernel, covered above; 20: the nearest-2x convolution in its sub-pixel form against the nine-tap
    # convolution of the upsampled image, and the refusal where the persistent kernel would not run)
    # (case 20 — the sub-pixel form through the entry point, a minute — runs under VSX_CPU_CHECK_FULL; the kernel itself: case 34 above)
    for case in [str(c) for c in range(13)] + ['14', '15', '16', '17', '18'] + (['20'] if os.environ.get('VSX_CPU_CHECK_FULL') else []):
        r = subprocess.run([exe, case], capture_output=True, text=True, timeout=600, env=env)
        print(r.stdout)
        assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    # round 5: the XCD block grid of the tile kernels (21: bit for bit like the linear walk, and actually chosen) and the residual
    # prefetch behind the last slab with its exact vmcnt counts (22) — the latter also with every DMA piece landing as late as
    # the kernel's own counted waits allow (an under-counted wait multiplies a slab that has not landed: the run fails)
    # 23: the persistent kernel's transposed store through the entry point, bit for bit like the tile kernels' (gemm_pp = 4)
    # 24 (round 6): the weight-stationary K = N = 320 kernel (gemm_ws320_kernel) — bias / residual / row statistics, bit for bit like the
    # tile kernels, with late-landing DMA: its counted waits include the stores of earlier blocks and two shortened counts in the
    # first two blocks (either one loosened by the size of its shortening fails this run)
    # 25: the same kernel over column slices (N = 640 / 960) with the folded LayerNorm and the gathered row vector
    for case, late in (('21', False), ('22', True), ('23', False), ('24', True), ('25', True)):
        r = subprocess.run([exe, case], capture_output=True, text=True, timeout=900,
                           env=dict(env, CPUHIP_DMA='late') if late else env)
        print('late DMA' if late else '', r.stdout)
        assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    for tile in ('1', '2', '3'):
        for case in ('0', '4', '8', '17'):      # (+res, GEGLU, LayerNorm fold, sub-pixel refusal; `make -C tools/cpu_check run` walks all seven)
            r = subprocess.run([exe, case], capture_output=True, text=True, timeout=600, env=dict(env, VSX_TUNE_TILE=tile))
            print('VSX_TUNE_TILE=' + tile, r.stdout)
            assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_normalisation_kernels_run_on_the_cpu(tmp_path):
    """csrc/norm.hip from its real source on the host: GroupNorm statistics / finalize / apply (4-D and 5-D scopes, the two-source
    concat with groups that straddle the sources, SiLU, C = 2560), LayerNorm (three vector widths, temporal positional encoding),
    the row statistics of the LayerNorm fold, plain and causal row softmax — against double precision
    (tools/cpu_check/check_norm.cpp)."""
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    exe = str(tmp_path / 'check_norm')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-Wno-unused-value', '-o', exe,
           os.path.join(src, 'check_norm.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_attention_kernels_run_on_the_cpu(tmp_path):
    """csrc/attention.hip on the host: the flash kernel (d = 40 / 64 / 80 / 160, ragged query and key tiles, K / V shared by two
    query batches), the temporal kernel with (site, head) problems packed into 32x32 MFMA tiles (16 and 24 frames, 4 frames with 8
    heads per tile), its long-clip form (16 x 64, 8 x 40 frames) and the VALU fallback, against a double-precision
    softmax(Q K^T) V (tools/cpu_check/check_attention.cpp; two textual rewrites of the source, attention_cpu.sed)."""
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    gen = tmp_path / 'gen'
    gen.mkdir()
    with open(gen / 'attention_cpu.hip', 'w') as f:
        r = subprocess.run(['sed', '-f', os.path.join(src, 'attention_cpu.sed'),
                            os.path.join(ROOT, 'videoswap_amd', 'csrc', 'attention.hip')], stdout=f, text=True)
    assert r.returncode == 0
    text = open(gen / 'attention_cpu.hip').read()
    assert 'cpuhip_dyn_lds' in text and text.count('wave_bar->arrive_and_wait(); const h8 vf') == 2      # both rewrites applied
    exe = str(tmp_path / 'check_attention')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', str(gen), '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-Wno-unused-value', '-Wno-division-by-zero',
           '-o', exe, os.path.join(src, 'check_attention.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]


def test_attention_backward_kernels_run_on_the_cpu(tmp_path):
    """csrc/attention_bwd.hip on the host: the dQ kernel and the dK / dV kernel of the flash-attention backward (d = 40 / 64 /
    80, ragged query and key tiles, text K / V shared by two images) and the delta kernel, against double-precision
    gradients of softmax(scale Q K^T) V (tools/cpu_check/check_attention_bwd.cpp)."""
    cxx = CXX if os.path.isfile(CXX) else shutil.which('clang++')
    src = os.path.join(ROOT, 'tools', 'cpu_check')
    exe = str(tmp_path / 'check_attention_bwd')
    cmd = [cxx, '-std=c++20', '-O1', '-pthread', '-I', src, '-I', os.path.join(ROOT, 'include'), '-I',
           os.path.join(ROOT, 'videoswap_amd', 'csrc'), '-Wno-unused-function', '-Wno-unused-value', '-Wno-division-by-zero',
           '-x', 'c++', '-o', exe, os.path.join(src, 'check_attention_bwd.cpp')]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(r.stdout)
    assert r.returncode == 0 and 'all checks passed' in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
