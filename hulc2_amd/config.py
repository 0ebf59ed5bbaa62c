"""The model configuration of the headline benchmark, as plain nested dicts.

Equals what Hydra composes from conf/cfg_low_level.yaml -> conf/model/calvin_hulc++.yaml with the overrides of
SURVEY.md §8d: `model/perceptual_encoder/rgb_static=default model.perceptual_encoder.rgb_static.input_height=200
datamodule.datasets.lang_dataset.load_lang_embeddings=true model/language_encoder=none`.  `_target_` strings are
the reference's own class paths (resolved to this package by hulc2_amd.compat).
"""
from .compat import Config


def default_model_config(gripper_control: bool = True, dropout_p: float = 0.1, static_hw=(200, 200)) -> Config:
    act7 = [1.0] * 7
    return Config.wrap({
        "_target_": "hulc2.models.hulc2.Hulc2",
        "_recursive_": False,
        "perceptual_encoder": {
            "_target_": "hulc2.models.perceptual_encoders.concat_encoders.ConcatEncoders",
            "_recursive_": False,
            "rgb_static": {
                "_target_": "hulc2.models.perceptual_encoders.vision_network.VisionNetwork",
                "input_width": static_hw[1], "input_height": static_hw[0], "activation_function": "ReLU", "dropout_vis_fc": 0.0,
                "l2_normalize_output": False, "visual_features": 64, "num_c": 3, "use_sinusoid": False, "spatial_softmax_temp": 1.0},
            "rgb_gripper": {
                "_target_": "hulc2.models.perceptual_encoders.vision_network_gripper.VisionNetwork",
                "input_width": 84, "input_height": 84, "activation_function": "ReLU", "dropout_vis_fc": 0.0,
                "l2_normalize_output": False, "visual_features": 64, "conv_encoder": "nature_cnn", "num_c": 3},
            "depth_static": None, "depth_gripper": None, "proprio": None, "tactile": None,
        },
        "plan_proposal": {
            "_target_": "hulc2.models.plan_encoders.plan_proposal_net.PlanProposalNetwork",
            "perceptual_features": None, "latent_goal_features": 32, "plan_features": None, "activation_function": "ReLU", "hidden_size": 2048},
        "plan_recognition": {
            "_target_": "hulc2.models.plan_encoders.plan_recognition_net.PlanRecognitionTransformersNetwork",
            "num_heads": 8, "num_layers": 2, "encoder_hidden_size": 2048, "fc_hidden_size": 4096, "in_features": None,
            "plan_features": None, "action_space": 7, "dropout_p": dropout_p, "encoder_normalize": False,
            "positional_normalize": False, "position_embedding": True, "max_position_embeddings": 32},
        "distribution": {"_target_": "hulc2.utils.distributions.Distribution", "dist": "discrete", "category_size": 32, "class_size": 32},
        "visual_goal": {
            "_target_": "hulc2.models.encoders.goal_encoders.VisualGoalEncoder", "in_features": None, "hidden_size": 2048,
            "latent_goal_features": 32, "l2_normalize_goal_embeddings": False, "activation_function": "ReLU"},
        "language_encoder": None,
        "language_goal": {
            "_target_": "hulc2.models.encoders.goal_encoders.LanguageGoalEncoder", "in_features": 384, "hidden_size": 2048,
            "latent_goal_features": 32, "l2_normalize_goal_embeddings": False, "activation_function": "ReLU", "word_dropout_p": 0.0},
        "action_decoder": {
            "_target_": "hulc2.models.decoders.logistic_decoder_rnn.LogisticDecoderRNN",
            "n_mixtures": 10, "hidden_size": 2048, "out_features": 7, "log_scale_min": -7.0, "act_max_bound": act7,
            "act_min_bound": [-v for v in act7], "dataset_dir": "", "load_action_bounds": False, "num_classes": 10,
            "latent_goal_features": 32, "plan_features": None, "perceptual_features": None, "gripper_alpha": 1.0,
            "perceptual_emb_slice": [64, 128], "policy_rnn_dropout_p": 0.0, "num_layers": 2, "rnn_model": "rnn_decoder",
            "gripper_control": gripper_control, "discrete_gripper": True},
        "optimizer": {"_target_": "torch.optim.Adam", "lr": 2e-4},
        "lr_scheduler": {"_target_": "transformers.get_constant_schedule"},
        "proj_vis_lang": {
            "_target_": "hulc2.models.auxiliary_loss_networks.proj_vis_lang.ProjVisLang", "im_dim": 4096, "lang_dim": 32,
            "output_dim": 32, "proj_lang": True},
        "kl_beta": 0.01, "kl_balancing_mix": 0.8, "replan_freq": 30, "use_clip_auxiliary_loss": True, "clip_auxiliary_loss_beta": 3.0,
    })


def real_world_model_config(dropout_p: float = 0.1) -> Config:
    """conf/cfg_low_level_rw.yaml -> conf/model/real_world_hulc++.yaml (BASELINE configs[3]): static camera through the frozen R3M trunk
    (conf/model/perceptual_encoder/rgb_static/r3m.yaml), the decoder sees the whole perceptual embedding and predicts world-frame actions
    (conf/model/action_decoder/logistic_decoder_rnn_real_world.yaml:15,19), no CLIP auxiliary loss (real_world_hulc++.yaml:12,20);
    language arrives as precomputed embeddings like the headline config (language_encoder = none)."""
    cfg = default_model_config(gripper_control=False, dropout_p=dropout_p)
    cfg["perceptual_encoder"]["rgb_static"] = Config.wrap({
        "_target_": "hulc2.models.perceptual_encoders.vision_r3m.VisionR3M", "visual_features": 64, "freeze_backbone": True,
        "resnet_model": "resnet18"})
    cfg["action_decoder"]["perceptual_emb_slice"] = [0, 128]
    cfg["proj_vis_lang"] = None
    cfg["use_clip_auxiliary_loss"] = False
    return cfg
