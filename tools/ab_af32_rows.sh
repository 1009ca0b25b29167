cd $GRAFT_REPO_ROOT
p() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3))"; }
for i in 1 2; do for r in 0 256 512 4096; do
CTI_AF32_ROWS=$r python bench.py --config c3 2>&1 | p "c3 af32_rows=$r"
CTI_AF32_ROWS=$r python bench.py --config c4 2>&1 | p "c4 af32_rows=$r"
done; done
