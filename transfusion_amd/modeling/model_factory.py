"""``get_fusion_model`` with the call shape of the reference's ``modeling/model_factory.py:73-115``: the build path
``runner/run_experiment.py:399-401`` uses, so the MI355X fusion block drops in behind the same run config."""
from __future__ import annotations

from transfusion_amd.modeling.cross_fusion.ego_fusion.cross_f_box_wrapper import CrossFusionBoxWrapper


def get_fusion_model(base_model, model_cfg, run_cfg, class_sizes):
    if run_cfg["narration_embeds"]["use"]:
        if run_cfg["narr_fusion"]["model"] == "cross_f":
            if run_cfg["experiment"] == "egonao":
                if run_cfg["narration_embeds"].get("res50_f", False) or run_cfg["narration_embeds"].get("slowfast_f_v", False):
                    raise NotImplementedError("VisLangFusionBoxWrapper (res50_f / slowfast_f_v) is off in the shipped configs; out of scope")
                if run_cfg["narr_fusion"]["share_encoders"]:
                    raise NotImplementedError("share_encoders: True (CrossFusionBoxWrapperShared) is broken in the reference "
                                              "(cross_f_box_wrapper.py:307) and out of scope")
                model = CrossFusionBoxWrapper(base_model, run_cfg["narr_fusion"], narr_embed_args=run_cfg["narration_embeds"]["args"],
                                              criterion=run_cfg["criterion"])
                # the reference hands run.precision to pl.Trainer (run_experiment.py:450); here it selects the encoders' arithmetic
                model.set_precision(run_cfg.get("precision", 32))
                return model
            raise NotImplementedError("only experiment: egonao reaches the fusion block (runner/utils/factories.py:11-20)")
        raise NotImplementedError(f'{run_cfg["narr_fusion"]["model"]=} is not implemented as fusion model.')
    return base_model
