"""Prompt layer of the hot path (string assembly only; mirrors the reference's call surface).

  * `Prompter`      -- utils/prompter.py:10-51 (template JSON under data/templates/, `generate_prompt`, `get_response`)
  * `Conversation`  -- demo.py:67-140 / test.py:150-198 (SeparatorStyle.SINGLE/TWO prompt assembly)
  * `report_prompt` -- the report-generation instruction with 32 x `<IMG>` (demo.py:262-266, vicuna_prompts.json:4)
"""
from __future__ import annotations

import dataclasses
import json
import os.path as osp
from enum import Enum, auto
from typing import Any, List, Optional, Union

_HERE = osp.dirname(osp.abspath(__file__))
N_IMG_TOKENS = 32
IMG_TOKEN = "<IMG>"


class Prompter(object):
    __slots__ = ("template", "_verbose")

    def __init__(self, template_name: str = "", verbose: bool = False):
        self._verbose = verbose
        if not template_name:
            template_name = "alpaca"           # the reference's default name; RaDialog itself passes "vicuna_v11"
        # the reference resolves data/templates/<name>.json relative to the CWD; fall back to the packaged copy
        candidates = [osp.join("data", "templates", f"{template_name}.json"),
                      osp.join(_HERE, "templates", f"{template_name}.json")]
        file_name = next((c for c in candidates if osp.exists(c)), None)
        if file_name is None:
            raise ValueError(f"Can't read {candidates[0]}")
        with open(file_name) as fp:
            self.template = json.load(fp)
        if self._verbose:
            print(f"Using prompt template {template_name}: {self.template['description']}")

    def generate_prompt(self, instruction: str, input: Union[None, str] = None, label: Union[None, str] = None) -> str:
        if input:
            res = self.template["prompt_input"].format(instruction=instruction, input=input)
        else:
            res = self.template["prompt_no_input"].format(instruction=instruction)
        if label:
            res = f"{res}{label}"
        if self._verbose:
            print(res)
        return res

    def get_response(self, output: str) -> str:
        return output.split(self.template["response_split"])[-1].strip()     # [-1]: multi-turn prompts


class SeparatorStyle(Enum):
    SINGLE = auto()
    TWO = auto()


@dataclasses.dataclass
class Conversation:
    """Keeps the conversation history and renders it to the Vicuna prompt."""
    system: str
    roles: List[str]
    messages: List[List[Optional[str]]]
    offset: int
    sep_style: SeparatorStyle = SeparatorStyle.SINGLE
    sep: str = "###"
    sep2: Optional[str] = None
    skip_next: bool = False
    conv_id: Any = None

    def get_prompt(self) -> str:
        if self.sep_style == SeparatorStyle.SINGLE:
            ret = self.system
            for role, message in self.messages:
                ret += self.sep + " " + role + ": " + message if message else self.sep + " " + role + ":"
            return ret
        if self.sep_style == SeparatorStyle.TWO:
            seps = [self.sep, self.sep2]
            ret = self.system + seps[0]
            for i, (role, message) in enumerate(self.messages):
                ret += role + ": " + message + seps[i % 2] if message else role + ":"
            return ret
        raise ValueError(f"Invalid style: {self.sep_style}")

    def clear(self):
        self.messages = []
        self.offset = 0
        self.skip_next = False

    def append_message(self, role, message):
        self.messages.append([role, message])

    def copy(self):
        return Conversation(system=self.system, roles=self.roles, messages=[[x, y] for x, y in self.messages],
                            offset=self.offset, sep_style=self.sep_style, sep=self.sep, sep2=self.sep2, conv_id=self.conv_id)

    def dict(self):
        return {"system": self.system, "roles": self.roles, "messages": self.messages, "offset": self.offset,
                "sep": self.sep, "sep2": self.sep2, "conv_id": self.conv_id}


VICUNA_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant."
                 "The assistant gives professional, detailed, and polite answers to the user's questions.")


def new_conversation() -> Conversation:
    """The conversation object demo.py:309-318 / test.py:118-127 build."""
    return Conversation(system=VICUNA_SYSTEM, roles=["USER", "ASSISTANT"], messages=[], offset=0,
                        sep_style=SeparatorStyle.TWO, sep=" ", sep2="</s>")


def report_prompt(findings: str) -> str:
    """demo.py:262-266: the report-generation instruction with the 32 image slots."""
    return (f"Image information: {IMG_TOKEN * N_IMG_TOKENS}. Predicted Findings: {findings}. You are to act as a radiologist and "
            "write the finding section of a chest x-ray radiology report for this X-ray image and the given predicted "
            "findings. Write in the style of a radiologist, write one fluent text without enumeration, be concise and "
            "don't provide explanations or reasons.")
