"""TEST INFRASTRUCTURE ONLY.  Builds tests/emu/_build/libes_emu_test.so: the kernel sources of embodiedscan_amd/csrc compiled
for x86 against the emulation shim (tests/emu/hip/hip_runtime.h) -- the same C ABI as libes_hip.so, operating on host
pointers, every launch executed by the fiber scheduler of emu_runtime.cpp.  The product never loads it.

The only source transformations: an occupancy attribute of the device compiler is dropped, `extern __shared__ T x[]` becomes a
pointer to the launch's dynamic-LDS buffer, the five `asm volatile("s_waitcnt ...")` statements of spconv.hip become calls that retire
the emulated LDS-DMA queue (x86 cannot assemble them) and the relative include of the public header is redirected.
    python tests/emu/build.py [file.hip ...]        (default: every source of the Makefile)"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'embodiedscan_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
CLANG = os.environ.get('ES_EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
DEFAULT = ('coords.hip', 'sort.hip', 'spconv.hip', 'rowops.hip', 'fusion.hip', 'targets.hip', 'losses.hip', 'optim.hip', 'data.hip',
           'predict.hip', 'dense.hip', 'occ.hip', 'transformer.hip', 'ground.hip', 'dconv.hip', 'halo.hip', 'imgwgrad.hip', 'imgconv.hip')


def transform(text):
    text = re.sub(r'asm volatile\("s_waitcnt vmcnt\(%0\)"\s*::\s*"n"\((\w+)\)\s*:\s*"memory"\);', r'ES_EMU_WAITCNT_VM(\1);', text)
    text = re.sub(r'asm volatile\("s_waitcnt vmcnt\((\d+)\)"\s*:::\s*"memory"\);', r'ES_EMU_WAITCNT_VM(\1);', text)
    text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(\d+\)"\s*:::\s*"memory"\);', 'ES_EMU_WAITCNT_NONE();', text)
    assert 'asm volatile' not in text, 'an inline-asm statement the emulation build does not know'
    text = re.sub(r'extern\s+__shared__\s+([\w ]+?)\s+(\w+)\[\];', r'\1* \2 = (\1*)emu::dyn_shared();', text)    # dynamic LDS
    text = re.sub(r'__attribute__\(\(amdgpu_waves_per_eu\(\d+\)\)\)', '', text)       # (an occupancy hint of the device compiler)
    return text.replace('#include "../../include/es_hip.h"', '#include "es_hip.h"')


def build(files=DEFAULT, force=False, lib_name='libes_emu_test.so'):
    """files: sources relative to embodiedscan_amd/csrc (sub-directories allowed: 'next/x.hip')"""
    import shutil
    if shutil.which(CLANG) is None and not os.path.exists(os.path.join(OUT, lib_name)):
        # a host without the ROCm clang (ES_EMU_CXX names another C++17 compiler): the emulator tests skip instead of erroring
        import pytest
        pytest.skip(f'{CLANG} not found (set ES_EMU_CXX to a C++17 compiler to build the emulated kernel library)')
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, lib_name)
    srcs = [os.path.join(CSRC, f) for f in files] + [os.path.join(CSRC, 'common.h'), os.path.join(ROOT, 'include', 'es_hip.h'),
                                                     os.path.join(HERE, 'hip', 'hip_runtime.h'), os.path.join(HERE, 'emu_runtime.cpp'),
                                                     os.path.join(HERE, 'selftest_kernels.cpp'),
                                                     os.path.abspath(__file__)]
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in srcs):
        return lib
    flags = ['-x', 'c++', '-std=c++17', '-O1', '-fPIC', '-w', '-I', HERE, '-I', CSRC, '-I', os.path.join(ROOT, 'include'),
             '-ffp-contract=off']
    jobs = []
    for f in files:
        gen = os.path.join(OUT, lib_name.replace('.so', '_') + f.replace('/', '_').replace('.hip', '_emu.cpp'))
        with open(os.path.join(CSRC, f)) as fh:
            text = transform(fh.read())
        with open(gen, 'w') as fh:
            fh.write(text)
        jobs.append((gen, gen.replace('.cpp', '.o')))
    jobs.append((os.path.join(HERE, 'selftest_kernels.cpp'), os.path.join(OUT, lib_name.replace('.so', '_selftest.o'))))
    rt = os.path.join(OUT, lib_name.replace('.so', '_runtime.o'))
    jobs.append((os.path.join(HERE, 'emu_runtime.cpp'), rt))
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:       # (the compiler processes run in parallel)
        list(ex.map(lambda j: subprocess.check_call([CLANG] + flags + ['-c', j[0], '-o', j[1]]), jobs))
    objs = [o for _, o in jobs[:-1]]
    subprocess.check_call([CLANG, '-shared', '-fPIC'] + objs + [rt, '-o', lib])
    return lib


if __name__ == '__main__':
    print(build(tuple(sys.argv[1:]) or DEFAULT, force=True))
