"""Downstream re-prompting on top of the generated reports -- the multi-turn callers of the hot path.

Reference: test.py:437-674 and downstream_tasks/{automated_correction,chexpert_classification_downstream}.py. After the report
pass (`preds_history` = the decoded "…USER: <report prompt> ASSISTANT:<report>" strings), three follow-up loops re-send the whole
conversation plus a new USER turn and decode a short answer:

  correction   (test.py:440-500)  batch 1, max_new_tokens 256: "Please adapt the report …" built from the false-positive /
               false-negative CheXpert labels of the first report (`get_correction_prompts`)
  binary QA    (test.py:548-570)  per study ONE batch of 14 prompts " Is there any <finding>?", max_new_tokens 10, answer = "yes" in it;
               "No Finding" is then derived from the other 13 columns
  findings QA  (test.py:610-650)  batch 5, max_new_tokens 30: "List all the findings in this report.", answer matched against the
               14 CheXpert names

Only the prompt construction, the generate calls (same arguments) and the answer parsing are mirrored; CheXbert labelling and the
sklearn metrics around them are not on the path. With `lang_model.reuse_prefix_kv = True` the decoder keeps the KV rows of the
token prefix a call shares with the previous one (rdx_generate_append) instead of re-running the whole conversation."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def _joined(names: List[str]) -> str:
    """'a, b and c' (automated_correction.py:15-20)."""
    return " and ".join(", ".join(names).rsplit(", ", 1))


def _fp_fn(col_names: Sequence[str], preds_row, labels_row) -> Tuple[List[str], List[str]]:
    fp = [col_names[i] for i, v in enumerate(preds_row * (1 - labels_row)) if v == 1 and col_names[i] != "No Finding"]
    fn = [col_names[i] for i, v in enumerate((1 - preds_row) * labels_row) if v == 1 and col_names[i] != "No Finding"]
    return fp, fn


def get_correction_prompts(preds_history: List[str], col_names: Sequence[str], chexpert_preds, chexpert_labels) -> List[str]:
    """automated_correction.py:3-35: one follow-up turn per study, in place (the caller's list is updated and returned)."""
    chexpert_preds, chexpert_labels = np.asarray(chexpert_preds), np.asarray(chexpert_labels)
    for idx in range(len(chexpert_preds)):
        fp, fn = _fp_fn(col_names, chexpert_preds[idx], chexpert_labels[idx])
        if fp and fn:
            corr = (f"Please adapt the report with the following corrections: Include {_joined(fn).lower()} and remove "
                    f"{_joined(fp).lower()}. Don't make other changes.")
        elif fp:
            corr = f"The patient does not have {_joined(fp).lower()}. Update the report. Don't make other changes."
        elif fn:
            corr = f"The patient also has {_joined(fn).lower()}, correct the report. Don't make other changes."
        else:
            corr = "KEEP_OLD"
        preds_history[idx] = preds_history[idx].replace("ASSISTANT:", "ASSISTANT: ") + "</s>USER: " + corr + " ASSISTANT:"
    return preds_history


def get_correction_labels(col_names: Sequence[str], chexpert_preds, chexpert_labels):
    """automated_correction.py:37-54."""
    chexpert_preds, chexpert_labels = np.asarray(chexpert_preds), np.asarray(chexpert_labels)
    both = [_fp_fn(col_names, p, l) for p, l in zip(chexpert_preds, chexpert_labels)]
    return [b[0] for b in both], [b[1] for b in both]


def get_chexpert_prompts_bin(preds_history: List, col_names: Sequence[str]) -> List[List[str]]:
    """chexpert_classification_downstream.py:1-17: 14 yes/no questions per study (in place, like the reference). Quirk kept: the
    reference re-applies `pred.replace("ASSISTANT:", "ASSISTANT: ")` inside the loop over the findings, and "ASSISTANT: " still
    contains "ASSISTANT:" -- so the k-th question's history carries k + 1 spaces behind every ASSISTANT: (the model was evaluated
    on exactly those strings)."""
    for idx, pred in enumerate(preds_history):
        questions = []
        for disease in col_names:
            pred = pred.replace("ASSISTANT:", "ASSISTANT: ")
            questions.append(pred + "</s>USER: " + " Is there any " + disease + "?" + " ASSISTANT:")
        preds_history[idx] = questions
    return preds_history


def get_chexpert_prompts_all(preds_history: List[str], col_names: Sequence[str]) -> List[str]:
    """chexpert_classification_downstream.py:19-27."""
    for idx, pred in enumerate(preds_history):
        preds_history[idx] = pred.replace("ASSISTANT:", "ASSISTANT: ") + "</s>USER: " + "List all the findings in this report." + " ASSISTANT:"
    return preds_history


def _generate(lang_model, tok, texts, dicom, max_new_tokens, num_beams=1):
    ids = tok.batch_encode_plus(list(texts), return_tensors="pt", padding=True)["input_ids"]
    out = lang_model.generate(input_ids=ids, return_dict_in_generate=True, output_scores=True, max_new_tokens=max_new_tokens,
                              dicom=dicom, num_beams=num_beams)
    return tok.batch_decode(out.sequences, skip_special_tokens=True)


def run_correction(lang_model, tok, correction_prompts: List[str], dicoms=None, num_beams: int = 1, max_new_tokens: int = 256) -> List[str]:
    """test.py:440-500: batch 1; studies whose prompt says KEEP_OLD keep their first report (test.py:483-486)."""
    preds = []
    for i, prompt in enumerate(correction_prompts):
        if "KEEP_OLD" in prompt:
            preds.append(prompt.split("ASSISTANT:")[-2].split("</s>")[0].strip())
            continue
        full = _generate(lang_model, tok, [prompt], None if dicoms is None else [dicoms[i]], max_new_tokens, num_beams)
        preds.append(full[0].split("ASSISTANT:")[-1].strip())
    return preds


def run_binary_qa(lang_model, tok, chexpert_prompts: List[List[str]], col_names: Sequence[str], dicoms=None, max_new_tokens: int = 10) -> np.ndarray:
    """test.py:548-590: one batch of 14 questions per study -> int[N, 14]; 'No Finding' = none of the other columns."""
    rows = []
    for i, questions in enumerate(chexpert_prompts):
        dic = None if dicoms is None else [dicoms[i]] * len(questions)
        preds = _generate(lang_model, tok, questions, dic, max_new_tokens)
        rows.append([1 if "yes" in p.split("ASSISTANT:")[-1].lower() else 0 for p in preds])
    out = np.array(rows, dtype=np.int64)
    nf = list(col_names).index("No Finding")
    rel = [j for j, c in enumerate(col_names) if c != "No Finding"]
    out[:, nf] = 1 - (out[:, rel].sum(axis=1) > 0)
    return out


def run_findings_qa(lang_model, tok, chexpert_prompts: List[str], col_names: Sequence[str], dicoms=None, batch_size: int = 5,
                    num_beams: int = 1, max_new_tokens: int = 30) -> np.ndarray:
    """test.py:610-650: batches of 5 -> int[N, 14] one-hot of the CheXpert names found in the answer."""
    rows = []
    for s in range(0, len(chexpert_prompts), batch_size):
        chunk = chexpert_prompts[s: s + batch_size]
        dic = None if dicoms is None else list(dicoms[s: s + batch_size])
        preds = [p.split("ASSISTANT:")[-1].lower() for p in _generate(lang_model, tok, chunk, dic, max_new_tokens, num_beams)]
        for ans in preds:
            rows.append([1 if label.lower() in ans else 0 for label in col_names])
    return np.array(rows, dtype=np.int64)
