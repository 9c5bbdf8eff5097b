"""Build libvsx.so (hand-written HIP kernels for gfx950) in-tree with hipcc.

    python -m videoswap_amd.build [--force] [--variant next]

hipcc cross-compiles for gfx950 without a GPU.  The shared library lands in videoswap_amd/lib/
(git-ignored, but it travels with the gpurun snapshot).

Variants: VARIANTS / VARIANT_EXTRA can name a development build (lib/libvsx_<name>.so with some sources swapped or
added; `VSX_LIB_VARIANT=<name>` makes `videoswap_amd._lib` load it) so that a candidate kernel can be A/B-ed on the GPU
without touching the measured library.  None exists at the moment: round 3 ran the round-2 candidates on hardware,
promoted the backward kernels and the strided all-to-all into libvsx.so and dropped the piece-schedule / packed-weight
fork of the persistent GEMM (no gain: profiles/r03_gemm_sched_ab_b2.txt, r03_gemm_bpack_ab_b2.txt).
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
# variant name -> {source file of SOURCES: replacement, relative to csrc/}
VARIANTS = {}
# variant name -> additional sources (relative to csrc/)
VARIANT_EXTRA = {}
SOURCES = ['api.cpp', 'comm.cpp', 'gemm.hip', 'gemm_pp.hip', 'norm.hip', 'attention.hip', 'attention_bwd.hip', 'elementwise.hip',
           'train.hip']
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-x', 'hip',
         '-I', os.path.join(ROOT, 'include'), '-I', CSRC, '-Wall', '-Wno-unused-function', '-Wno-division-by-zero',
         # MFMA results stay in architectural VGPRs: the attention kernels post-process every score on the VALU, and the
         # default AGPR form cost 159 v_accvgpr_read/write moves per 64-key tile (measured in the ISA)
         '-mllvm', '-amdgpu-mfma-vgpr-form']


def lib_path(variant=None):
    return LIB if not variant else os.path.join(LIBDIR, f'libvsx_{variant}.so')


def _digest(variant=None):
    """sha256 over the kernel sources, the public header and the compiler flags.  Location-independent: the flags
    are hashed with the checkout path stripped (the gpurun snapshot lives under another root than the build tree).
    A variant hashes its replacement files in place of the ones they replace (and its name)."""
    h = hashlib.sha256()
    swap = VARIANTS[variant] if variant else {}
    names = [n for n in sorted(os.listdir(CSRC)) if n.endswith(('.hip', '.cpp', '.h'))]
    for name in names + ['../../include/vsx.h']:
        with open(os.path.join(CSRC, swap.get(name, name)), 'rb') as f:
            h.update(name.encode() + b'\0' + f.read())
    h.update(' '.join(f.replace(ROOT, '<root>') for f in FLAGS).encode())
    if variant:
        for name in VARIANT_EXTRA.get(variant, []):
            with open(os.path.join(CSRC, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
        h.update(b'\0variant ' + variant.encode())
    return h.hexdigest()


def source_digest(variant=None):
    """Digest of everything libvsx.so is built from; embedded in the binary (vsx_source_digest)."""
    return _digest(variant)


_MARKER = b'@vsx-source-digest:'


def built_digest(variant=None):
    """Digest embedded in the existing libvsx.so, or None (missing / predates the marker).  Read from the file's bytes,
    not through dlopen: ctypes never unloads, so probing a stale library in this process would pin the old image
    under its name and the library rebuilt right after would still answer with the old digest."""
    path = lib_path(variant)
    if not os.path.exists(path):
        return None
    with open(path, 'rb') as f:
        blob = f.read()
    i = blob.find(_MARKER)
    if i < 0:
        return None
    j = blob.find(b'\0', i)
    return blob[i + len(_MARKER):j].decode(errors='replace')


def _compile(src, digest, variant=None):
    objdir = OBJDIR if not variant else os.path.join(LIBDIR, f'obj_{variant}')
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, src.replace('/', '_') + '.o')
    extra = [f'-DVSX_SOURCE_DIGEST="{digest}"'] if src == 'api.cpp' else []
    path = os.path.join(CSRC, (VARIANTS[variant] if variant else {}).get(src, src))
    cmd = [HIPCC] + FLAGS + extra + ['-c', path, '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {src}:\n{r.stdout}\n{r.stderr}')
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build(force=False, verbose=True, variant=None):
    os.makedirs(OBJDIR, exist_ok=True)
    lib = lib_path(variant)
    digest = _digest(variant)
    # the digest lives INSIDE the binary (no side-car stamp file that git could update without the .so)
    if not force and built_digest(variant) == digest:
        if verbose:
            print(f'[vsx] {lib} is up to date')
        return lib
    if not os.path.exists(HIPCC):
        raise RuntimeError(f'{HIPCC} not found: cannot build {os.path.basename(lib)}')
    sources = SOURCES + (VARIANT_EXTRA.get(variant, []) if variant else [])
    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(lambda src: _compile(src, digest, variant), sources))
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl']
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
    if verbose:
        print(f'[vsx] built {lib}')
    return lib


if __name__ == '__main__':
    _variant = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else None
    build(force='--force' in sys.argv, variant=_variant)
