#!/bin/bash
# tools/run_lg4x.sh [max compared calls] [check|device|host] : the reference's LG4X mixture analysis with the class trees mirrored on the device
# (oracle/glue_driver.c, check mode); prints the GLUE_DRIVER report line
b=/tmp/lg4x_run; rm -rf $b; mkdir -p $b/examples/lg4x $b/run
cp /root/repo/tests/golden/lg4x/* $b/examples/lg4x/; cp /root/repo/tests/golden/examples_proteic.phy $b/examples/proteic
cd $b/run && GLUE_MODE=${2:-check} GLUE_DEVICE_PMAT=${3:-0} GLUE_MAX_MIXT=${1:-0} /root/repo/oracle/_ref/phyml_glue_driver -- --xml=../examples/lg4x/lg4x_check.xml 2>&1 | grep -E "GLUE_DRIVER|Log-likelihood"
