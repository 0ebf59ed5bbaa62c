"""Parameter / buffer names and shapes of the checkpoint contract.

The names are what PyTorch derives from the reference's attribute names (hulc2/models/hulc2.py:71-99 and
the leaf modules); a checkpoint written by either implementation loads into the other.  Used to build
synthetic state_dicts without instantiating modules and to test that the product modules expose
exactly these keys.
"""
from collections import OrderedDict
from typing import Dict, Tuple


def trainable_shapes(static_hw=(200, 200), hidden=2048, plan=1024, n_mix=10, act_dims=6, fc_hidden=4096,
                     d_model=128, ff=2048, n_layers=2, max_pos=32, goal=32, lang_in=384,
                     decoder_in=1120) -> "OrderedDict[str, Tuple[int, ...]]":
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()

    def lin(name, o, i):
        s[name + ".weight"] = (o, i)
        s[name + ".bias"] = (o,)

    def ln(name, d):
        s[name + ".weight"] = (d,)
        s[name + ".bias"] = (d,)

    for cam in ("rgb_static_encoder", "rgb_gripper_encoder"):
        p = f"perceptual_encoder.{cam}."
        s[p + "conv_model.0.weight"], s[p + "conv_model.0.bias"] = (32, 3, 8, 8), (32,)
        s[p + "conv_model.2.weight"], s[p + "conv_model.2.bias"] = (64, 32, 4, 4), (64,)
        s[p + "conv_model.4.weight"], s[p + "conv_model.4.bias"] = (64, 64, 3, 3), (64,)
        if cam == "rgb_gripper_encoder":
            lin(p + "conv_model.7", 128, 64 * 7 * 7)
        lin(p + "fc1.0", 512, 128)
        lin(p + "fc2", 64, 512)
        ln(p + "ln", 64)
    # plan proposal (prior)
    lin("plan_proposal.fc_model.0", hidden, d_model + goal)
    for i in (2, 4, 6):
        lin(f"plan_proposal.fc_model.{i}", hidden, hidden)
    lin("plan_proposal.fc_state.0", plan, hidden)
    # plan recognition (posterior) transformer
    s["plan_recognition.position_embeddings.weight"] = (max_pos, d_model)
    ln("plan_recognition.layernorm", d_model)
    for l in range(n_layers):
        p = f"plan_recognition.transformer_encoder.layers.{l}."
        s[p + "self_attn.in_proj_weight"] = (3 * d_model, d_model)
        s[p + "self_attn.in_proj_bias"] = (3 * d_model,)
        lin(p + "self_attn.out_proj", d_model, d_model)
        lin(p + "linear1", ff, d_model)
        lin(p + "linear2", d_model, ff)
        ln(p + "norm1", d_model)
        ln(p + "norm2", d_model)
    lin("plan_recognition.fc", fc_hidden, d_model)
    lin("plan_recognition.fc_state.0", plan, fc_hidden)
    # goal encoders
    lin("visual_goal.mlp.0", hidden, d_model)
    lin("visual_goal.mlp.2", hidden, hidden)
    lin("visual_goal.mlp.4", goal, hidden)
    ln("visual_goal.ln", goal)
    lin("language_goal.mlp.1", hidden, lang_in)
    lin("language_goal.mlp.3", hidden, hidden)
    lin("language_goal.mlp.5", goal, hidden)
    ln("language_goal.ln", goal)
    # action decoder
    for l, i in ((0, decoder_in), (1, hidden)):
        s[f"action_decoder.rnn.weight_ih_l{l}"] = (hidden, i)
        s[f"action_decoder.rnn.weight_hh_l{l}"] = (hidden, hidden)
        s[f"action_decoder.rnn.bias_ih_l{l}"] = (hidden,)
        s[f"action_decoder.rnn.bias_hh_l{l}"] = (hidden,)
    for h in ("mean_fc", "log_scale_fc", "prob_fc"):
        lin("action_decoder." + h, act_dims * n_mix, hidden)
    lin("action_decoder.gripper_fc", 2, hidden)
    # CLIP auxiliary projections
    lin("proj_vis_lang.mlp_im.0", 128, fc_hidden)
    lin("proj_vis_lang.mlp_im.2", goal, 128)
    lin("proj_vis_lang.mlp_lang.0", 128, goal)
    lin("proj_vis_lang.mlp_lang.2", goal, 128)
    s["logit_scale"] = ()
    return s


def buffer_shapes(n_mix=10, act_dims=6) -> Dict[str, Tuple[int, ...]]:
    return {
        "perceptual_encoder.rgb_static_encoder.spatial_softmax.x_map": (441,),
        "perceptual_encoder.rgb_static_encoder.spatial_softmax.y_map": (441,),
        "perceptual_encoder.rgb_static_encoder.spatial_softmax.temperature": (1,),
        "action_decoder.one_hot_embedding_eye": (n_mix, n_mix),
        "action_decoder.ones": (1, 1, n_mix),
        "action_decoder.gripper_bounds": (2,),
        "action_decoder.action_max_bound": (1, 1, act_dims, n_mix),
        "action_decoder.action_min_bound": (1, 1, act_dims, n_mix),
    }


def num_trainable(**kw) -> int:
    n = 0
    for shp in trainable_shapes(**kw).values():
        k = 1
        for d in shp:
            k *= d
        n += k
    return n
