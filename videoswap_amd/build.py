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
import re
import shutil
import subprocess
import sys
import tempfile
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


LLVM_BIN = os.environ.get('VSX_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
# kernels whose inner loops count their own `vmcnt` entries (LDS-DMA pieces, residual ring): a scratch reload is one more
# VMEM operation in that queue, so these must compile without spills and without a private segment (gemm_pp.hip:33-37)
# (matched on the MANGLED names: gemm_kernelILi256E... = gemm_kernel<256, ...>)
NO_SCRATCH = ('gemm_pp_kernel', 'gemm_ws320_kernel', 'flash_attn_kernel', 'gemm_kernelILi256E')


def parse_kernel_notes(text):
    """AMDGPU metadata note (llvm-readelf --notes) -> {mangled kernel name: {vgpr_count, agpr_count, sgpr_count,
    vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size, group_segment_fixed_size}}."""
    keys = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count',
            'private_segment_fixed_size', 'group_segment_fixed_size')
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r'^\s+(-\s+)?\.(\w+):\s*(\S.*)?$', line)
        if not m:
            continue
        if m.group(1) and re.match(r'^  - ', line):        # a new entry of amdhsa.kernels (the nested .args entries sit deeper)
            cur = {}
        if cur is None:
            continue
        key, val = m.group(2), (m.group(3) or '').strip()
        if key == 'name' and re.match(r'^    \.name:', line):
            out[val] = cur
        elif key in keys and re.match(r'^  (- |  )\.', line):
            cur[key] = int(val)
    return out


def kernel_resources(lib=None):
    """Register / scratch figures of every kernel in the built library, from the code objects' own metadata notes
    (llvm-objdump --offloading extracts the gfx950 bundles, llvm-readelf --notes prints them)."""
    lib = lib or LIB
    objdump, readelf = os.path.join(LLVM_BIN, 'llvm-objdump'), os.path.join(LLVM_BIN, 'llvm-readelf')
    tmp = tempfile.mkdtemp(prefix='vsx_co_')
    try:
        copy = os.path.join(tmp, 'lib.so')
        shutil.copy(lib, copy)
        r = subprocess.run([objdump, '--offloading', copy], capture_output=True, text=True, cwd=tmp)
        if r.returncode != 0:
            raise RuntimeError(f'llvm-objdump --offloading failed:\n{r.stderr}')
        res = {}
        for name in sorted(os.listdir(tmp)):
            if 'gfx950' not in name:
                continue
            r = subprocess.run([readelf, '--notes', os.path.join(tmp, name)], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f'llvm-readelf --notes failed on {name}:\n{r.stderr}')
            res.update(parse_kernel_notes(r.stdout))
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def demangle(names):
    """pretty names for messages only (binutils' c++filt when there is one; the checks match mangled names)"""
    filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
    out = []
    if filt and names:
        r = subprocess.run([filt], input='\n'.join(names), capture_output=True, text=True)
        out = r.stdout.splitlines() if r.returncode == 0 else []
    return dict(zip(names, out if len(out) == len(names) else names))


def scratch_offenders(resources, patterns=NO_SCRATCH):
    """[(demangled name, vgpr spills, private segment bytes)] of the kernels matching `patterns` that spill or own a
    private segment."""
    pretty = demangle(list(resources))
    bad = []
    for name, r in resources.items():
        if not any(p in name for p in patterns):
            continue
        if r.get('vgpr_spill_count', 0) or r.get('private_segment_fixed_size', 0):
            bad.append((pretty[name], r.get('vgpr_spill_count', 0), r.get('private_segment_fixed_size', 0)))
    return sorted(bad)


def check_no_scratch(lib=None):
    """Post-link check: the hand-scheduled kernels must not touch scratch."""
    res = kernel_resources(lib)
    if not res:
        raise RuntimeError('no kernel metadata found in ' + (lib or LIB))
    bad = scratch_offenders(res)
    if bad:
        raise RuntimeError('kernels that must not use scratch do:\n' + '\n'.join(
            f'  {n}: {s} VGPRs spilled, {b} B private segment' for n, s, b in bad))
    return res


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
    if os.environ.get('VSX_ALLOW_SCRATCH') != '1':     # development builds may look at a spilling candidate; the product may not
        try:
            check_no_scratch(lib)
        except RuntimeError:
            os.replace(lib, lib + '.rejected')         # never leave a library behind that the check refused
            raise
    if verbose:
        print(f'[vsx] built {lib}')
    return lib


def build_from_git(name, rev, verbose=True):
    """lib/libvsx_<name>.so from the kernel sources of git revision `rev` (A/B of two BUILDS inside one process on the GPU box:
    tools/gemm_ab.py --libs product,<name>).  The old sources are extracted into lib/src_<name>/ (git-ignored with the rest of
    lib/); nothing of them is tracked, and the product never loads such a library (VSX_LIB_VARIANT / the tools only)."""
    src = os.path.join(LIBDIR, f'src_{name}')
    inc = os.path.join(src, 'include')
    shutil.rmtree(src, ignore_errors=True)
    os.makedirs(inc)
    ls = subprocess.run(['git', '-C', ROOT, 'ls-tree', '--name-only', rev, 'videoswap_amd/csrc/'], capture_output=True, text=True, check=True)
    files = [f for f in ls.stdout.split() if f.endswith(('.hip', '.cpp', '.h'))] + ['include/vsx.h']
    for f in files:
        blob = subprocess.run(['git', '-C', ROOT, 'show', f'{rev}:{f}'], capture_output=True, check=True).stdout
        with open(os.path.join(inc if f.startswith('include/') else src, os.path.basename(f)), 'wb') as out:
            out.write(blob)
    objdir = os.path.join(LIBDIR, f'obj_{name}')
    os.makedirs(objdir, exist_ok=True)
    flags = [f if f not in (os.path.join(ROOT, 'include'), CSRC) else (inc if f.endswith('include') else src) for f in FLAGS]
    sources = [n for n in SOURCES if os.path.exists(os.path.join(src, n))]

    def one(n):
        obj = os.path.join(objdir, n + '.o')
        extra = [f'-DVSX_SOURCE_DIGEST="git-{rev[:12]}"'] if n == 'api.cpp' else []
        r = subprocess.run([HIPCC] + flags + extra + ['-c', os.path.join(src, n), '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {n} of {rev}:\n{r.stderr}')
        return obj
    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(one, sources))
    lib = lib_path(name)
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr}')
    if verbose:
        print(f'[vsx] built {lib} from {rev}')
    return lib


def build_with_defines(name, defines, verbose=True):
    """lib/libvsx_<name>.so from the CURRENT sources with extra -D flags (ablation builds for tools/gemm_ab.py --libs: what an
    epilogue costs without its stores / its activation ...).  Never the product: no digest, no scratch check, results may be wrong."""
    objdir = os.path.join(LIBDIR, f'obj_{name}')
    os.makedirs(objdir, exist_ok=True)

    def one(n):
        obj = os.path.join(objdir, n + '.o')
        extra = [f'-DVSX_SOURCE_DIGEST="defines-{name}"'] if n == 'api.cpp' else []
        r = subprocess.run([HIPCC] + FLAGS + list(defines) + extra + ['-c', os.path.join(CSRC, n), '-o', obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {n} ({defines}):\n{r.stderr}')
        return obj
    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(one, SOURCES))
    lib = lib_path(name)
    r = subprocess.run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + ['-ldl'], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'link failed:\n{r.stderr}')
    if verbose:
        print(f'[vsx] built {lib} with {" ".join(defines)}')
    return lib


if __name__ == '__main__':
    if '--define' in sys.argv:                         # --define NAME -DFLAG[=v] [-DFLAG2 ...]
        _i = sys.argv.index('--define')
        build_with_defines(sys.argv[_i + 1], [a for a in sys.argv[_i + 2:] if a.startswith('-D')])
        sys.exit(0)
    if '--from-git' in sys.argv:                       # --from-git NAME REV
        _i = sys.argv.index('--from-git')
        build_from_git(sys.argv[_i + 1], sys.argv[_i + 2])
        sys.exit(0)
    _variant = sys.argv[sys.argv.index('--variant') + 1] if '--variant' in sys.argv else None
    if '--resources' in sys.argv:                      # register / scratch table of the built library
        _res = kernel_resources(lib_path(_variant))
        _names = demangle(list(_res))
        for _k, _r in sorted(_res.items(), key=lambda kv: _names[kv[0]]):
            print(f"{_r.get('vgpr_count', 0):4d} v {_r.get('agpr_count', 0):4d} a {_r.get('vgpr_spill_count', 0):3d} spilled "
                  f"{_r.get('private_segment_fixed_size', 0):5d} B scratch {_r.get('group_segment_fixed_size', 0):7d} B LDS  {_names[_k]}")
        sys.exit(0)
    build(force='--force' in sys.argv, variant=_variant)
