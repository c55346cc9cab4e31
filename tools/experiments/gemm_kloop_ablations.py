import subprocess, sys, os
src = open('/root/repo/transfusion_amd/csrc/gemm_bf16.hip').read()
def variant(name, edits):
    s = src
    for old, new in edits:
        assert s.count(old) == 1, (name, old[:60], s.count(old))
        s = s.replace(old, new)
    path = f'/root/repo/transfusion_amd/csrc/_gemm_{name}.hip'
    open(path, 'w').write(s)
    obj = f'/root/repo/transfusion_amd/csrc/_obj/_gemm_{name}.o'
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-munsafe-fp-atomics', '-Wno-unused-result', '-c', path, '-o', obj], capture_output=True, text=True)
    os.remove(path)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    objs = [obj] + [f'/root/repo/transfusion_amd/csrc/_obj/{n}.o' for n in ('attn_bf16', 'attn_x3', 'rowops', 'heads', 'comm', 'tf_api')]
    r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', f'/root/repo/transfusion_amd/lib/libtfusion_{name}.so'] + objs + ['-ldl'], capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    print('built', name)
W_READ = ('''    bf16x8 wf[4], xf[MF];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(slw + j * 1024);
#pragma unroll
    for (int j = 0; j < MF; ++j) xf[j] = *(const bf16x8*)(slx + j * 1024);''')
HOIST = ('''  int cur = 0;                                                     // ring slot of step i
  for (int i = 0; i < nk; ++i) {''', '''  int cur = 0;                                                     // ring slot of step i
  bf16x8 wf[4], xf[MF];
  for (int i = 0; i < nk; ++i) {''')
# (d) neither operand read after step 0
variant('noreads', [HOIST, (W_READ, '''    if (i == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(slw + j * 1024);
#pragma unroll
    for (int j = 0; j < MF; ++j) xf[j] = *(const bf16x8*)(slx + j * 1024);
    }''')])
# (b) no DMA inside the loop
variant('nodma', [('''    stage(cur == 0 ? NSLOT - 1 : cur - 1, min(i + DIST, nk - 1));''', '''    if (i < 0) stage(cur == 0 ? NSLOT - 1 : cur - 1, min(i + DIST, nk - 1));''')])
# (c) no barrier / wait in the loop
variant('nobar', [('''    asm volatile("s_waitcnt vmcnt(%0)\\n\\ts_barrier" ::"n"((DIST - 1) * PER_WAVE) : "memory");''', '''    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((DIST - 1) * PER_WAVE) : "memory");''')])
# (e) one MFMA in four (matrix work / 4)
variant('mfma4', [('''          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }''', '''          if (ni == 0) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }''')])
import sys
sys.argv = ['x']
exec(open('/root/repo/tools/experiments/gemm_kloop_ablations.py').read().split("W_READ = (")[0])
# DMA pattern of 8 rows x 128 B per instruction (whole 128-B lines) instead of 16 rows x 64 B: timing only (garbage operands)
LINE128 = [('''    const int r = (isW ? j - NA : j) * 16 + (lane >> 2);
    const int c = (lane & 3) ^ swz<32>(r);''', '''    const int r = (isW ? j - NA : j) * 8 + (lane >> 3);
    const int c = (lane & 7);'''),
 ('''      const unsigned char* base = (isW ? Wp : Ap) + (size_t)kstep * (BIG_BK * 2);''',
  '''      const unsigned char* base = (isW ? Wp : Ap) + (size_t)(kstep % (nk0 / 2)) * (BIG_BK * 4);''')]
MF4 = ('''          acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }''', '''          if (ni == 0) acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
      }''')
variant('line128', LINE128)
variant('line128mf4', LINE128 + [MF4])
