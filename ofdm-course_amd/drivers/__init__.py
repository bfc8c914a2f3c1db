"""Replays of the reference's six driver scripts over this package's API.

The `.m` drivers cannot run anywhere in this pipeline (no MATLAB), so each `taskN.run()` repeats the
call order and parameter names of its script -- `Main_model.m`, `Main_model_Task_2..5.m`,
`Task5_part2.m` -- with every numeric step going through the HIP library.  Plots, image I/O and the
PAPR/CCDF study are left out; the payload is seeded synthetic bits.  Each run
returns the tables the script prints / plots (BER, MER, MSE(SNR), NMSE / BER per comb) as a dict;
`python -m ofdm_course_amd.drivers.task3 --json out.json` writes them as JSON.
`sweep_ber` is the multi-GPU Monte-Carlo sweep of the fused chain (one process per GPU, `torch.distributed`).
"""
from . import common, task1, task2, task3, task4, task5, task5_part2  # noqa: F401
