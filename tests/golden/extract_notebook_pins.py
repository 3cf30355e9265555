"""Transcribe the stored OUTPUTS of the reference's tutorial notebook into a JSON fixture.

The reference (pasqal-io/pulser-diff) cannot be imported in the build container
(pyqtorch / pulser / qutip absent) and its tests hold no static vectors
(SURVEY.md section 8c).  The only static numerical pins are the outputs stored in
``docs/basic_usage.ipynb``.  This script copies those printed NUMBERS (data, not
source) into ``notebook_pins.json``:

  KA-1  DP5_SE, 4 qubits, sampling_rate 0.1: evaluation times (160), <sum Z>(t)
        (160 values, 4 decimals), printed amplitudes of the first/last states.
  KA-2..4  KRYLOV_SE, 2 qubits, sampling_rate 0.5: initial <sum Z>(T) prints.
  KA-5  Adam loss traces printed by the optimisation loops (6 decimals).

Run (only where /root/reference exists):  python tests/golden/extract_notebook_pins.py
"""
import json
import re
import sys
from pathlib import Path

NB = Path("/root/reference/docs/basic_usage.ipynb")
OUT = Path(__file__).with_name("notebook_pins.json")

_FLOAT = r"[-+]?\d+\.\d+(?:e[-+]?\d+)?"


def _out_text(cell):
    chunks = []
    for o in cell.get("outputs", []):
        if "text" in o:
            chunks.append("".join(o["text"]))
        elif "data" in o and "text/plain" in o["data"]:
            chunks.append("".join(o["data"]["text/plain"]))
    return "\n".join(chunks)


def _tensor_floats(txt):
    body = txt[txt.index("tensor(") :]
    body = body[: body.index("dtype")]
    return [float(v) for v in re.findall(_FLOAT, body)]


def _loss_trace(txt):
    init = float(re.search(r"Initial expectation value: tensor\((" + _FLOAT + r")", txt).group(1))
    final = float(re.search(r"Optimized expectation value: tensor\((" + _FLOAT + r")", txt).group(1))
    losses = [float(v) for v in re.findall(r"\[\d+\] loss: (" + _FLOAT + r")", txt)]
    return {"initial_expectation": init, "optimized_expectation": final, "losses": losses}


def _complex_rows(txt):
    rows = re.findall(r"\[\s*(" + _FLOAT + r")([-+]\d+\.\d+(?:e[-+]?\d+)?)j\]", txt)
    return [[float(a), float(b)] for a, b in rows]


def main():
    nb = json.loads(NB.read_text())
    cells = nb["cells"]
    pins = {"source": "docs/basic_usage.ipynb stored outputs (pulser-diff @ 2025-06-14)"}

    # KA-1: cell printing evaluation times + wavefunctions, and the cell echoing exp_val
    txt14 = _out_text(cells[14])
    times_txt = txt14[: txt14.index("Wavefunctions")]
    pins["ka1_eval_times"] = _tensor_floats(times_txt)
    # printed state rows: 5 printed time slices (first 3, last 2... as shown), each 6 rows (3 head, 3 tail)
    pins["ka1_state_rows"] = _complex_rows(txt14[txt14.index("Wavefunctions") :])
    pins["ka1_sum_z"] = _tensor_floats(_out_text(cells[25]))

    # KA-2..KA-5: optimisation loop prints
    pins["ka2_pulse_opt"] = _loss_trace(_out_text(cells[44]))
    pins["ka3_register_opt"] = _loss_trace(_out_text(cells[55]))
    pins["ka_duration_opt"] = _loss_trace(_out_text(cells[65]))   # needs tanh-envelope model: not a pin
    pins["ka4_shape_opt"] = _loss_trace(_out_text(cells[81]))
    pins["ka_noisy_opt"] = _loss_trace(_out_text(cells[89]))      # DP5_ME: out of scope, kept for the record

    assert len(pins["ka1_eval_times"]) == 160, len(pins["ka1_eval_times"])
    assert len(pins["ka1_sum_z"]) == 160, len(pins["ka1_sum_z"])
    OUT.write_text(json.dumps(pins, indent=1))
    print("wrote", OUT, {k: (len(v) if isinstance(v, list) else "obj") for k, v in pins.items()})


if __name__ == "__main__":
    sys.exit(main())
