"""pixel_aware_gyro_aided_klt_feature_tracker_amd.runtime_env: process-level ROCm runtime defaults, set on import unless the
environment already says otherwise."""
import importlib
import os

from pixel_aware_gyro_aided_klt_feature_tracker_amd import runtime_env


def test_the_package_import_put_the_default_in_force():
    assert os.environ.get("DEBUG_CLR_GRAPH_PACKET_CAPTURE") is not None
    assert runtime_env.IN_FORCE["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"]


def test_a_value_of_the_environment_wins(monkeypatch):
    monkeypatch.setenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "1")
    assert runtime_env.apply() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "1"}
    assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "1"


def test_unset_is_filled_in_and_the_opt_out_leaves_it_alone(monkeypatch):
    monkeypatch.delenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE", raising=False)
    monkeypatch.setenv("PAGK_KEEP_RUNTIME_ENV", "1")
    assert runtime_env.apply() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": None}
    assert "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ
    monkeypatch.delenv("PAGK_KEEP_RUNTIME_ENV")
    assert runtime_env.apply() == {"DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}
    assert os.environ["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] == "0"
