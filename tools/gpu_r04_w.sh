#!/bin/bash
# forward wave priorities (GSR_FWD_PRIO): same-box A/B on the three workloads
cd $GRAFT_REPO_ROOT
O=gpurun_out
{
timeout 300 python tools/ab_variants.py --smoke --no-extra-configs base prio1=GSR_FWD_PRIO=1 prio2=GSR_FWD_PRIO=2 base2 prio1b=GSR_FWD_PRIO=1 prio2b=GSR_FWD_PRIO=2
timeout 300 python tools/ab_variants.py --no-extra-configs --s0 0.05 base prio1=GSR_FWD_PRIO=1 prio2=GSR_FWD_PRIO=2
timeout 300 python tools/ab_variants.py --no-extra-configs --steps 50 --gaussians 6000000 base prio1=GSR_FWD_PRIO=1 prio2=GSR_FWD_PRIO=2
} > $O/r04_w_prio.txt 2>&1
cat $O/r04_w_prio.txt
