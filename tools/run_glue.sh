cd /root/repo/tests/golden
for ex in nucleic proteic; do
  d=/tmp/g_$ex; mkdir -p $d; cp examples_$ex.phy $d/$ex
done
cd /tmp/g_nucleic
for mode in check device; do GLUE_MODE=$mode /root/repo/oracle/_ref/phyml_glue_driver --gtr-rr 1,2.5,0.8,1.2,3.0,1 -- -i nucleic -d nt -m GTR -f 0.3,0.2,0.2,0.3 -c 4 -a 0.8 -s SPR -o tl -b 0 --r_seed 1 2>&1 | grep GLUE_DRIVER | cut -c1-420; done
GLUE_MODE=device GLUE_DEVICE_PMAT=1 /root/repo/oracle/_ref/phyml_glue_driver --gtr-rr 1,2.5,0.8,1.2,3.0,1 -- -i nucleic -d nt -m GTR -f 0.3,0.2,0.2,0.3 -c 4 -a 0.8 -s SPR -o tl -b 0 --r_seed 1 2>&1 | grep GLUE_DRIVER | cut -c1-420
cd /tmp/g_proteic
for mode in device; do GLUE_MODE=$mode /root/repo/oracle/_ref/phyml_glue_driver -- -i proteic -d aa -m LG -f m -c 4 -a 1.0 -s SPR -o tl -b 0 --r_seed 1 2>&1 | grep GLUE_DRIVER | cut -c1-420; done
