python tools/encw_fine.py encwnoload 32 2>&1 | tail -6
python tools/encw_fine.py encwnpf2 32 2>&1 | tail -6
