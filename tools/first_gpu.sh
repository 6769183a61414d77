set -e
python -c "import __graft_entry__ as g; g.build()"
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -40
