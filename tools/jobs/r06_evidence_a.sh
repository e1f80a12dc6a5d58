#!/bin/bash
# round 6 evidence, part 1c: the Model-A legs again with 1 024 rotating index buffers (64 sat inside the Infinity Cache)
cd $GRAFT_REPO_ROOT
for p in roofline value A256; do ONLY=$p bash tools/jobs/r06_evidence.sh 2>&1 | grep "done"; done
PMC_PART=a2 PMC_KEYS="fused_m2_A256" bash tools/jobs/r06_evidence_pmc.sh 2>&1 | tail -2
