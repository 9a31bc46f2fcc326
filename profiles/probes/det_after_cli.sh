run() { echo "== $1: $(env $1 python tools/probes/det_after_cli.py 2>&1 | grep '^0 ' | python -c "import sys,ast; l=sys.stdin.read(); rows=ast.literal_eval(l[2:]); print([round(r[1],3) for r in rows])")"; }
for v in "$@"; do run "$v"; done
