# Runs the C++ demo executable (the reference main.cpp equivalent) on the golden Epoch_001/002 scans: bash tools/demo_check.sh on the GPU box
set -e
cd $GRAFT_REPO_ROOT
D=$(mktemp -d)
printf "string FolderFilePath1: %s\nstring FolderFilePath2: %s\nbool isSetResSVsize (yes-1, no-0): 1\nfloat PCres1 (m): 0.005\nfloat PCres2 (m): 0.005\nfloat SVsize1 (m): 0.05\nfloat SVsize2 (m): 0.05\nbool isSetDTinit (yes-1, no-0): 1\nfloat DTinit (m): 0.05\nfloat DTmin (m): 0.004\nbool isVisual (yes-1, no-0): 0" tests/golden/inputs/Epoch_001.pcd tests/golden/inputs/Epoch_002.pcd > $D/cfg.txt
./piecewise-icp_amd/pwicp_demo pair $D/cfg.txt $D/out_ | tail -3
ls $D
head -6 $D/out_TransMatrix.txt
