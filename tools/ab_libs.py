"""Dev tool: A/B two builds of libhbvx.so on the same box.

    AB_CFGS="hourly cfg3" python tools/ab_libs.py libhbvx_A.so libhbvx_B.so

runs tools/bench_configs.py with each library (separate processes, three interleaved rounds) and prints
(config, ms per step, forward ms, backward ms) per round; the libraries are looked up in hydrodl2_amd/csrc/."""
import os, sys, shutil, subprocess, json
root=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs=sys.argv[1:]
cfgs=os.environ.get('AB_CFGS','cfg2').split()
csrc=os.path.join(root,'hydrodl2_amd','csrc')
shutil.copy(os.path.join(csrc,'libhbvx.so'), os.path.join(csrc,'libhbvx_base.so'))
res={l:[] for l in libs}
for rnd in range(3):
    for l in libs:
        shutil.copy(os.path.join(csrc,l), os.path.join(csrc,'libhbvx.so'))
        outs=[o for o in subprocess.run([sys.executable, os.path.join(root,'tools','bench_configs.py')]+cfgs,capture_output=True,text=True).stdout.strip().split('\n') if o.startswith('{')]
        res[l].append([(json.loads(o)['config'], json.loads(o)['ms_per_step']) + tuple(v for k, v in json.loads(o)['kernel_ms'].items() if k.endswith(('forward', 'backward')) and 'route' not in k) + ((json.loads(o)['kernel_ms'],) if os.environ.get('AB_ALL') else ()) for o in outs])
shutil.copy(os.path.join(csrc,'libhbvx_base.so'), os.path.join(csrc,'libhbvx.so'))
for l,v in res.items(): print(l, v)
