"""Builds libmgp.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m multiagent_gnn_policies_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  Objects go to csrc/build/ (git-ignored); the
shared library lands next to this file so it travels with the repo snapshot to the GPU box.
"""
import hashlib
import os
import re
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
OBJ_DIR = os.path.join(CSRC, 'build')
LIB_PATH = os.path.join(PKG_DIR, 'libmgp.so')
STAMP = os.path.join(OBJ_DIR, 'libmgp.srchash')

ARCH = 'gfx950'
COMMON_FLAGS = ['--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
# flock.hip must reproduce numpy's op-by-op fp64 rounding: no fused multiply-add contraction
# -fno-slp-vectorize (the resident rollout kernels): under plain -O3 the compiler packs adjacent scalar fp32 adds / multiplies of
# the pair tests into v_pk_*_f32 behind register shuffles and s_nops; measured on the headline build: one-step launches 15.6 ->
# 14.0 us with the packing off, longer launches 1 % (profiles/r05_rollout_ab.txt)
_RO = ['-ffp-contract=off', '-fno-slp-vectorize']
PER_FILE_FLAGS = {'flock.hip': ['-ffp-contract=off'], 'rollout.hip': _RO,
                  'sparse_sim.hip': ['-ffp-contract=off'], 'rollout_wide.hip': _RO,
                  'rollout_w128.hip': _RO, 'rollout_f32ref.hip': _RO, 'rollout_t512.hip': _RO, 'rollout_w128x2.hip': _RO, 'rollout_w128xd.hip': _RO}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def source_hash():
    h = hashlib.sha256()
    names = sources() + sorted(f for f in os.listdir(CSRC) if f.endswith('.h'))
    names.append(os.path.join('..', '..', 'include', 'mgp.h'))
    for n in names:
        with open(os.path.join(CSRC, n), 'rb') as f:
            h.update(n.encode())
            h.update(f.read())
    h.update(' '.join(COMMON_FLAGS).encode())
    for n in sorted(PER_FILE_FLAGS):                     # e.g. -ffp-contract=off decides the fp64 rounding of the simulator
        h.update((n + ':' + ' '.join(PER_FILE_FLAGS[n])).encode())
    return h.hexdigest()


def hipcc_path():
    p = shutil.which('hipcc')
    if p is None and os.path.exists('/opt/rocm/bin/hipcc'):
        p = '/opt/rocm/bin/hipcc'
    return p


def is_current():
    if not (os.path.exists(LIB_PATH) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as f:
        return f.read().strip() == source_hash()


def build(force=False, verbose=True):
    """Compile every .hip under csrc/ for gfx950 and link libmgp.so.  Returns the library path.
    Serialised across processes by an exclusive lock on csrc/build/.lock: under torchrun every rank imports the package
    at once, and concurrent builds would race on the same object files and on libmgp.so.tmp."""
    if not force and is_current():
        return LIB_PATH
    import fcntl
    os.makedirs(OBJ_DIR, exist_ok=True)
    with open(os.path.join(OBJ_DIR, '.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and is_current():               # another process built it while we waited
                return LIB_PATH
            return _build_locked(verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


_INCLUDE_RE = re.compile(rb'^[ \t]*#[ \t]*include[ \t]*"([^"]+)"', re.M)


def _tu_bytes(src, seen=None):
    """The bytes a translation unit is built from besides the headers: its own text plus, recursively, every file it
    pulls in with #include "..." (rollout_wide.hip / rollout_w128.hip are wrappers around rollout.hip: editing
    rollout.hip must rebuild all three objects)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(os.path.join(CSRC, src))
    if path in seen or not os.path.exists(path):
        return b''
    seen.add(path)
    with open(path, 'rb') as f:
        text = f.read()
    out = src.encode() + b'\0' + text
    for inc in _INCLUDE_RE.findall(text):
        out += _tu_bytes(os.path.join(os.path.dirname(src), inc.decode()), seen)
    return out


def _build_locked(verbose, force=False):
    hipcc = hipcc_path()
    if hipcc is None:
        raise RuntimeError("hipcc not found: cannot build libmgp.so (ROCm toolchain required)")
    os.makedirs(OBJ_DIR, exist_ok=True)
    objs = []
    procs = []
    # per-object stamps: a translation unit is recompiled only when it, a header or its flags changed
    hdr = hashlib.sha256()
    for n in sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'mgp.h')]:
        with open(os.path.join(CSRC, n), 'rb') as f:
            hdr.update(n.encode())
            hdr.update(f.read())
    for src in sources():
        obj = os.path.join(OBJ_DIR, src[:-4] + '.o')
        flags = COMMON_FLAGS + PER_FILE_FLAGS.get(src, [])
        key = hashlib.sha256(hdr.digest() + _tu_bytes(src) + ' '.join(flags).encode()).hexdigest()
        stamp = obj + '.key'
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read().strip() == key:
            continue
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print('[mgp build]', ' '.join(cmd), flush=True)
        procs.append((src, stamp, key, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, stamp, key, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors='replace')))
        if verbose and out.strip():
            print(out.decode(errors='replace'))
        with open(stamp, 'w') as f:
            f.write(key)
    tmp = LIB_PATH + '.tmp.%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', tmp] + objs
    if verbose:
        print('[mgp build]', ' '.join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors='replace'))
    os.replace(tmp, LIB_PATH)
    with open(STAMP, 'w') as f:
        f.write(source_hash())
    return LIB_PATH


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
