import os

from transfusion_amd.runner.config import load_yaml, update_config

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_update_config_derives_fusion_keys(monkeypatch):
    monkeypatch.setenv("CODE", ROOT)
    for name, d in (("ego_nao_res50_ego4d.yml", 712), ("ego_nao_res50_ego4dv2.yml", 896)):
        cfg = update_config(load_yaml(os.path.join(ROOT, "transfusion_amd", "runner", "configs", name)))
        run = cfg["run"]
        assert run["narr_fusion"]["args"]["input_f_size"] == d           # run_experiment.py:99-100
        assert run["narr_fusion"]["type"] == "cross_transformer" and run["narr_fusion"]["args"]["num_layers"] == [4, 4, 4, 4]
        assert run["narration_embeds"]["args"]["text_pooling"] == "slowfast"
        assert run["experiment"] == "egonao"
        assert run["narr_fusion"]["patch_h"] == [4, 4, 2, 1]
