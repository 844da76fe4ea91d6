for v in "GRIDPF_DENSE=1" "GRIDPF_FORCE_GENERIC=1" "GRIDPF_IPW=1" "GRIDPF_IPW=4" "GRIDPF_DENSE=1 GRIDPF_DENSE64=1"; do
echo "== $v"
env $v python -m pytest tests -m gpu -x -q 2>&1 | tail -2
done
