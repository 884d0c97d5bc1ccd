cd $GRAFT_REPO_ROOT
bash tools/pmc_block.sh 0 pmc_r02f_l0enc g1 g2 g3 g4
