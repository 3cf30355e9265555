"""Transcribe the stored OUTPUTS of the reference's tutorial notebook into a JSON fixture.

The reference (pasqal-io/pulser-diff) cannot be imported in the build container
(pyqtorch / pulser / qutip absent) and its tests hold no static vectors
(SURVEY.md section 8c).  The only static numerical pins are the outputs stored in
``docs/basic_usage.ipynb``, ``docs/state_preparation.ipynb`` and ``docs/gate_optimization.ipynb``.  This script
copies those printed NUMBERS (data, not source) into ``notebook_pins.json``:

  KA-1  DP5_SE, 4 qubits, sampling_rate 0.1: evaluation times (160), <sum Z>(t)
        (160 values, 4 decimals), printed amplitudes of the first/last states.
  KA-2..4  KRYLOV_SE, 2 qubits, sampling_rate 0.5: initial <sum Z>(T) prints.
  KA-5  Adam loss traces printed by the optimisation loops (6 decimals).
  KA-6  state_preparation.ipynb: the optimised 30+30 pulse-shape parameters printed in full (4 decimals), the best loss
        (1 - fidelity, 16 digits) and the printed final fidelity (6 qubits, Rydberg level 60, DP5_SE, rate 0.05).
  KA-7  gate_optimization.ipynb part 1: 24 optimised constant-pulse parameters (amplitude, detuning, phase x 8), best
        loss and printed gate fidelity (2 qubits, all 4 basis states as one batch).
  KA-8  gate_optimization.ipynb part 2: 20+20 pulse-shape parameters, best loss, gate fidelity (4 qubits, 16 columns).
  KA-7's run starts from fixed parameters (all 5.0), so its first printed loss (6 decimals) is a forward pin as well.
  The random initial parameters of the other runs are not stored, but the FINAL parameters are, and the forward pass at
  those parameters is deterministic: loss(final parameters) must reproduce the printed numbers.

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


def _named_parameters(txt):
    """'<module>.<name>\nParameter containing:\ntensor(<values>, requires_grad=True)' blocks -> {name: [values]}."""
    out = {}
    for name, body in re.findall(r"(\w+\.\w+)\nParameter containing:\ntensor\((.*?),\s+requires_grad=True\)", txt, flags=re.S):
        out[name.split(".")[-1]] = [float(v) for v in re.findall(r"[-+]?\d+\.\d*(?:e[-+]?\d+)?", body)]
    return out


def _optimised_run(cells, loss_cell, final_cell, what, init_cell=None):
    loss_txt, final_txt = _out_text(cells[loss_cell]), _out_text(cells[final_cell])
    best = re.search(r"Best loss: (" + _FLOAT + r") after (\d+) epochs", loss_txt)
    fid = re.search(what + r" fidelity: (" + _FLOAT + r")%", final_txt)
    first = re.search(r"\[t=0\]loss: (" + _FLOAT + r")", loss_txt)
    trace = {int(k): float(v) for k, v in re.findall(r"\[t=(\d+)\]loss: (" + _FLOAT + r")", loss_txt)}
    extra = {"loss_trace": trace}
    if init_cell is not None:  # the shaped-pulse runs print their (random) INITIAL parameters too: the trace becomes reproducible
        extra["initial_parameters"] = _named_parameters(_out_text(cells[init_cell]))
    return {**extra, "first_loss": float(first.group(1)), "best_loss": float(best.group(1)), "best_epoch": int(best.group(2)), "printed_fidelity_percent": float(fid.group(1)),
            "parameters": _named_parameters(final_txt)}


def main():
    nb = json.loads(NB.read_text())
    cells = nb["cells"]
    pins = {"source": "docs/{basic_usage,state_preparation,gate_optimization}.ipynb stored outputs (pulser-diff @ 2025-06-14)"}

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

    # KA-6..KA-8: final parameters + best loss + printed fidelity of the two optimal-control notebooks
    sp = json.loads(NB.with_name("state_preparation.ipynb").read_text())["cells"]
    go = json.loads(NB.with_name("gate_optimization.ipynb").read_text())["cells"]
    pins["ka6_state_preparation"] = _optimised_run(sp, 10, 12, "State", init_cell=8)
    pins["ka7_gate_constant_pulses"] = _optimised_run(go, 13, 16, "Gate")
    pins["ka8_gate_pulse_shape"] = _optimised_run(go, 25, 28, "Gate", init_cell=23)
    assert [len(v) for v in pins["ka6_state_preparation"]["parameters"].values()] == [30, 30]
    assert len(pins["ka7_gate_constant_pulses"]["parameters"]) == 24
    assert [len(v) for v in pins["ka8_gate_pulse_shape"]["parameters"].values()] == [20, 20]

    assert len(pins["ka1_eval_times"]) == 160, len(pins["ka1_eval_times"])
    assert len(pins["ka1_sum_z"]) == 160, len(pins["ka1_sum_z"])
    OUT.write_text(json.dumps(pins, indent=1))
    print("wrote", OUT, {k: (len(v) if isinstance(v, list) else "obj") for k, v in pins.items()})


if __name__ == "__main__":
    sys.exit(main())
