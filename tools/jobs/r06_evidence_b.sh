#!/bin/bash
# round 6 evidence, part 1b: the Model-B legs again with 32 rotating index buffers in the single-configuration runs (as on the default line)
cd $GRAFT_REPO_ROOT
ONLY=B1024 bash tools/jobs/r06_evidence.sh 2>&1 | tail -8
PMC_PART=b2 PMC_KEYS="fused_h_B1024_bf16" bash tools/jobs/r06_evidence_pmc.sh 2>&1 | tail -3
