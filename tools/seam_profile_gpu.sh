cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf /tmp/lkref && mkdir -p /tmp/lkref && tar -C /tmp/lkref -xzf .stage/lkref.tar.gz
export LK_REFERENCE_ROOT=/tmp/lkref
R="$PWD"
export PYTHONPATH="$R/oracle/shims:/tmp/lkref/src:$R"
export LD_PRELOAD=/usr/lib/x86_64-linux-gnu/libstdc++.so.6
/opt/conda/bin/python3.9 -W ignore tools/seam_profile.py 20000 2>&1 | grep -v "not evenly\|Method has been" | cut -c1-150
