import os, sys, shutil, subprocess, json
sys.path.insert(0,'/root/repo')
# A/B: run bench_configs with each library build (separate processes), several rounds interleaved
import itertools
root=os.environ.get('GRAFT_REPO_ROOT','/root/repo')
libs=sys.argv[1:]
cfgs=os.environ.get('AB_CFGS','cfg2').split()
csrc=os.path.join(root,'hydrodl2_amd','csrc')
shutil.copy(os.path.join(csrc,'libhbvx.so'), os.path.join(csrc,'libhbvx_base.so'))
res={l:[] for l in libs}
for rnd in range(3):
    for l in libs:
        shutil.copy(os.path.join(csrc,l), os.path.join(csrc,'libhbvx.so'))
        outs=[o for o in subprocess.run([sys.executable, os.path.join(root,'tools','bench_configs.py')]+cfgs,capture_output=True,text=True).stdout.strip().split('\n') if o.startswith('{')]
        res[l].append([(json.loads(o)['config'], json.loads(o)['kernel_ms']['hbvx_forward'], json.loads(o)['kernel_ms']['hbvx_backward']) for o in outs])
shutil.copy(os.path.join(csrc,'libhbvx_base.so'), os.path.join(csrc,'libhbvx.so'))
for l,v in res.items(): print(l, v)
