"""The package keeps the reference's public names and call signatures.

``tests/fixtures/reference_api_surface.json`` is the reference's surface as parsed from its
sources by ``tools/api_surface_check.py --write`` (modules, ``__all__``, classes with
methods and fields, functions, parameter names and order); nothing of the reference is
imported or run.
"""

import copy
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.fixture(scope="module")
def tool():
    spec = importlib.util.spec_from_file_location(
        "api_surface_check", os.path.join(ROOT, "tools", "api_surface_check.py"))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


@pytest.fixture(scope="module")
def surface(tool):
    with open(tool.FIXTURE) as fh:
        return json.load(fh)


def test_every_reference_name_and_signature_is_served(tool, surface):
    assert len(surface) >= 40 and sum(len(m["classes"]) for m in surface.values()) >= 80
    assert tool.compare(surface) == []


def test_the_check_notices_missing_names_and_changed_signatures(tool, surface):
    broken = copy.deepcopy(surface)
    broken["server"]["all"].append("NoSuchAggregator")
    fedavg = broken["server.aggregator.fedavg"]["classes"]["FedAvgAggregator"]
    fedavg["methods"]["aggregate"]["pos"].reverse()
    fedavg["methods"]["no_such_method"] = {"pos": ["self"], "kw": [], "var": False, "varkw": False,
                                           "defaults": {}}
    broken["utils.dates"]["functions"]["get_current_time"]["kw"].append("no_such_keyword")
    broken["privacy.constants"]["values"]["MAX_EPSILON"] = 11.0
    broken["trainer.base"]["classes"]["TrainingConfig"]["values"]["max_batches"] = 7
    broken["no.such.module"] = {"all": None, "classes": {}, "functions": {}, "names": []}
    problems = "\n".join(tool.compare(broken))
    assert "NoSuchAggregator" in problems
    assert "FedAvgAggregator.aggregate" in problems
    assert "no_such_method" in problems
    assert "get_current_time" in problems
    assert "no.such.module" in problems
    assert "MAX_EPSILON" in problems
    assert "TrainingConfig.max_batches" in problems


def test_parser_reads_signatures_without_importing(tool, tmp_path):
    src = tmp_path / "m.py"
    src.write_text(
        "raise SystemExit('never executed')\n"
        "__all__ = ['fn', 'Cls']\n"
        "LIMIT: int = 3\n"
        "T = TypeVar('T')\n"
        "DEFAULT = 4\n"
        "def fn(a, b=1, *, c=2, **rest): ...\n"
        "class Cls:\n"
        "    x: int = 0\n"
        "    def m(self, p): ...\n"
        "    @property\n"
        "    def q(self): ...\n"
        "    def _hidden(self): ...\n"
    )
    info = tool.parse_module(str(src))
    assert info["all"] == ["fn", "Cls"] and info["names"] == ["LIMIT", "DEFAULT"]
    assert info["functions"]["fn"] == {"pos": ["a", "b"], "kw": ["c"], "var": False, "varkw": True,
                                       "defaults": {"b": 1, "c": 2}}
    assert info["values"] == {"LIMIT": 3, "DEFAULT": 4}
    assert info["classes"]["Cls"]["fields"] == ["x"]
    assert info["classes"]["Cls"]["values"] == {"x": 0}
    assert info["classes"]["Cls"]["methods"] == {
        "m": {"pos": ["self", "p"], "kw": [], "var": False, "varkw": False, "defaults": {}},
        "q": None}
