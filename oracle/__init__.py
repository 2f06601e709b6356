"""CPU oracle for the jellyfish MSM/NTT hot path -- TEST INFRASTRUCTURE ONLY (see pyref.py)."""
