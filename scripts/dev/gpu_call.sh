cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for bk in 16 32 16 32; do OAT_LIN_BK=$bk timeout 200 python scripts/dev/text_alone.py 2>&1 | tail -1 | sed "s/^/BK=$bk /"; done
