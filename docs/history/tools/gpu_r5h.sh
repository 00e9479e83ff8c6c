#!/bin/bash
export TMPDIR=/tmp
cd /tmp && rm -rf fb && mkdir fb && cd fb
cp $GRAFT_REPO_ROOT/tests/golden/examples_nucleic.phy nucleic
python - <<PY
import json,os,subprocess,sys
root=os.environ["GRAFT_REPO_ROOT"]
e=json.load(open(root+"/tests/golden/search_expected.json"))["search_nucleic_spr"]
env=dict(os.environ,GLUE_MODE="check",GLUE_FIRST_BAD="1")
r=subprocess.run([root+"/oracle/_ref/phyml_glue_driver"]+e["driver_opts"]+["--","-i","nucleic"]+e["phyml_args"],env=env,stdout=subprocess.PIPE,stderr=subprocess.STDOUT,text=True)
for l in r.stdout.splitlines():
    if "GLUE_FIRST_BAD" in l or "GLUE_TRACE" in l: print(l[:200])
PY
