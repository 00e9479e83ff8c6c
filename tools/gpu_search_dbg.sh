#!/bin/bash
export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
python - <<'PY'
import os, subprocess, sys, tempfile, re
ROOT=os.getcwd(); sys.path.insert(0, ROOT)
from phyml_amd import synth
n,P=150,20000
tmp=tempfile.mkdtemp(prefix="search_")
tree=synth.random_tree(n,11,0.02,0.15); st=synth.simulate_states(tree,P,4,11)
synth.write_phylip(os.path.join(tmp,"ali.phy"),tree.names,synth.states_to_chars(st,4))
args=["--gtr-rr","1,2.5,0.8,1.2,3.0,1","--","-i","ali.phy","-d","nt","-m","GTR","-f","0.3,0.2,0.2,0.3","-c","4","-a","0.8","-s","SPR","-o","tl","-b","0","--r_seed","1","--no_colalias"]
for res in ("1",):
    r=subprocess.run([os.path.join(ROOT,"oracle","_ref","phyml_glue_driver")]+args,cwd=tmp,env=dict(os.environ,GLUE_MODE="device",GLUE_DEVICE_PMAT="1",PHYHIP_RESIDENT=res,PHYHIP_RESIDENT_STATS="1"),stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
    for line in r.stdout.splitlines():
        if "resident" in line or "GLUE_DRIVER" in line: print(line[:400])
PY
