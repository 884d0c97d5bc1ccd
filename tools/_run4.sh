cd $GRAFT_REPO_ROOT
bash tools/pmc_block.sh 1 pmc_r02k_l1enc g1 g2 g3 g4
