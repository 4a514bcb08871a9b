"""Model configuration for the BEVBert cross-modal hot path.

Restates the constants of the reference's JSON configs (configs/r2r_model.json,
configs/rxr_model.json, configs/rvr_model.json) as a plain attribute bag; the
reference builds a transformers ``PretrainedConfig`` from those files
(pretrain_src/train_r2r.py:102-113), here nothing depends on transformers.
"""
import json


class BevBertConfig:
    # configs/r2r_model.json (R2R pre-training defaults)
    _DEFAULTS = dict(
        hidden_size=768,
        num_attention_heads=12,
        intermediate_size=3072,
        hidden_act="gelu",
        hidden_dropout_prob=0.1,
        attention_probs_dropout_prob=0.1,
        pred_head_dropout_prob=0.1,
        layer_norm_eps=1e-12,
        max_position_embeddings=512,
        type_vocab_size=2,
        vocab_size=30522,
        image_feat_size=512,
        angle_feat_size=4,
        obj_feat_size=0,
        obj_prob_size=0,
        depth_feat_size=0,     # continuous-environment fork: DD-PPO depth features per view (bevbert_ce r2r_model_config_dep.json)
        loc_feat_size=None,    # None -> angle_feat_size + 3 (vilmodel.py:471); the CE fork uses angle_feat_size alone
        nav_type_vocab=3,      # 0 non-navigable, 1 navigable, 2 object (vilmodel.py:480-481); CE fork: 2
        num_l_layers=9,
        num_x_layers=4,
        num_pano_layers=2,
        max_action_steps=100,
        update_lang_bert=True,
        use_lang2visn_attn=True,
        graph_sprels=True,
        glocal_fuse=True,
        bev_dim=21,
        bev_res=0.5,           # pretrain_src/model/pretrain_cmt.py:17
        grid_feat_size=768,    # hard-coded nn.Linear(768, hidden): vilmodel.py:577
        grid_hw=14,            # 14x14 ViT patch grid: pretrain_cmt.py:22-23
        grid_views=12,
        sem_classes=40,        # pretrain_cmt.py:68
        feat_dropout=0.4,
        output_attentions=False,
        # fine-tune only (map_nav_src/models/vlnbert_init.py:57-76)
        fix_lang_embedding=False,
        fix_pano_embedding=False,
        fix_local_branch=False,
        # task selection (pretrain_src/train_r2r.py:108-112)
        pretrain_tasks=("mlm", "sap", "masksem"),
        sem_pred_token="cattn",
    )

    def __init__(self, **kw):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        for k, v in kw.items():
            setattr(self, k, v)
        self.pretrain_tasks = set(self.pretrain_tasks)
        if self.loc_feat_size is None:
            self.loc_feat_size = self.angle_feat_size + 3

    @classmethod
    def adopt(cls, config):
        """Complete a foreign configuration object.  The reference hands its models a transformers ``PretrainedConfig``
        built from configs/*_model.json plus ``pretrain_tasks`` (set or list) and ``sem_pred_token``
        (pretrain_src/train_r2r.py:102-113); those files do not carry the constants the reference hard-codes in its
        model code (BEV resolution 0.5 m, the 14x14x12 ViT grid, 768-d grid features, 40 semantic classes:
        pretrain_src/model/pretrain_cmt.py:16-32,68; vilmodel.py:577).  Any attribute bag / dict is accepted: known
        keys are taken from it, everything else falls back to ``_DEFAULTS``; the caller's object is not modified."""
        if isinstance(config, cls):
            return config
        if isinstance(config, dict):
            get, has = config.__getitem__, config.__contains__
        else:
            get, has = (lambda k: getattr(config, k)), (lambda k: hasattr(config, k))
        kw = {k: get(k) for k in cls._DEFAULTS if has(k) and get(k) is not None}
        if isinstance(kw.get("pretrain_tasks"), str):
            kw["pretrain_tasks"] = kw["pretrain_tasks"].split(".")[::2]      # "mlm.5.sap.5.masksem.1" (opts.task_ratio)
        return cls(**kw)

    @classmethod
    def from_json_file(cls, path, **kw):
        with open(path) as f:
            d = json.load(f)
        d.update(kw)
        return cls(**d)

    @classmethod
    def rxr(cls, **kw):
        # configs/rxr_model.json: xlm-roberta vocabulary
        return cls(vocab_size=250002, max_position_embeddings=514, **kw)

    @classmethod
    def reverie(cls, **kw):
        # configs/rvr_model.json: ImageNet ViT features (768) + object tokens; tasks of scripts/pt_rvr.bash
        d = dict(image_feat_size=768, obj_feat_size=768, obj_prob_size=1000, pretrain_tasks=("mlm", "mrc", "sap", "og"))
        d.update(kw)
        return cls(**d)

    @classmethod
    def ce(cls, **kw):
        """bevbert_ce/pretrain/run_pt/r2r_model_config_dep.json + pretrain_cmt.py:16-17: 11x11 BEV at 1 m, a depth-feature
        branch in the panorama embedding, 4-d location features, 2 nav types, tasks mlm + sap, no semantics."""
        d = dict(bev_dim=11, bev_res=1.0, depth_feat_size=128, loc_feat_size=4, nav_type_vocab=2, sem_classes=0,
                 pretrain_tasks=("mlm", "sap"))
        d.update(kw)
        return cls(**d)

    @classmethod
    def tiny(cls, **kw):
        """Small depth/vocab for golden fixtures and CPU tests (widths are kept:
        768 is hard-coded in the reference's BEV embedding)."""
        d = dict(num_l_layers=2, num_x_layers=2, num_pano_layers=1, vocab_size=1200,
                 max_position_embeddings=128)
        d.update(kw)
        return cls(**d)

    def to_dict(self):
        d = {k: getattr(self, k) for k in self._DEFAULTS}
        d["pretrain_tasks"] = sorted(self.pretrain_tasks)
        return d
