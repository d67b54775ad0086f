for l in -1 -3 -10 -30; do echo "LAM0=$l"; COPRA_OPTIONS=ric_lam0=$l python tools/riccati_mfma_profile.py 16384 2>&1 | grep -E "no profile|iters mean"; done
python -m pytest tests/test_gpu_parity.py -q -x -k "config5 or riccati or beyond" 2>&1 | tail -3
