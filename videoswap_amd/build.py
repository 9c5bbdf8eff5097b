"""Build libvsx.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python -m videoswap_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The shared library lands in videoswap_amd/lib/
(git-ignored, but it travels with the gpurun snapshot).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(LIBDIR, 'obj')
LIB = os.path.join(LIBDIR, 'libvsx.so')
SOURCES = ['api.cpp', 'comm.cpp', 'gemm.hip', 'gemm_pp.hip', 'norm.hip', 'attention.hip', 'elementwise.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
         '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-Wall', '-Wno-unused-function', '-Wno-division-by-zero',
         # MFMA results stay in architectural VGPRs: the attention kernels post-process every score on the VALU, and the
         # default AGPR form cost 159 v_accvgpr_read/write moves per 64-key tile (measured in the ISA)
         '-mllvm', '-amdgpu-mfma-vgpr-form']


def _digest():
    """sha256 over the kernel sources, the public header and the compiler flags.  Location-independent: the flags
    are hashed with the checkout path stripped (the gpurun snapshot lives under another root than the build tree)."""
    h = hashlib.sha256()
    names = [n for n in sorted(os.listdir(CSRC)) if n.endswith(('.hip', '.cpp', '.h'))]
    for name in names + ['../../include/vsx.h']:
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    h.update(' '.join(f.replace(ROOT, '<root>') for f in FLAGS).encode())
    return h.hexdigest()


def source_digest():
    """Digest of everything libvsx.so is built from; embedded in the binary (vsx_source_digest)."""
    return _digest()


def built_digest():
    """Digest embedded in the existing libvsx.so, or None (missing / predates the symbol)."""
    if not os.path.exists(LIB):
        return None
    import ctypes
    try:
        fn = ctypes.CDLL(LIB).vsx_source_digest
    except (OSError, AttributeError):
        return None
    fn.restype = ctypes.c_char_p
    return fn().decode()


def _compile(src, digest):
    obj = os.path.join(OBJDIR, src + '.o')
    extra = [f'-DVSX_SOURCE_DIGEST="{digest}"'] if src == 'api.cpp' else []
    cmd = [HIPCC] + FLAGS + extra + ['-c', os.path.join(CSRC, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    digest = _digest()
    # the digest lives INSIDE the binary (no side-car stamp file that git could update without the .so)
    if not force and built_digest() == digest:
        if verbose:
            print(f'[vsx] {LIB} is up to date')
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f'{HIPCC} not found: cannot build libvsx.so')
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(lambda src: _compile(src, digest), SOURCES))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[vsx] built {LIB}')
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
