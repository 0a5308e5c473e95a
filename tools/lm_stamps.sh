cd $GRAFT_REPO_ROOT
SOICP_EXTRA_CXXFLAGS=-DSO_LM_STAMPS python -m superodom_amd.build --force > /dev/null 2>&1
SOICP_ABLATE=128 SOICP_LM_STAMPS=1 python tools/eval_stamps.py 2>&1 | grep -v "^\s*$" | head -40
