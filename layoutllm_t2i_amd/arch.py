"""Static description of the layout-conditioned UNet: block plan and parameter table.

The reference builds its UNet imperatively inside ``UNetModel.__init__``
(GLIGEN/ldm/modules/diffusionmodules/openaimodel.py:234-391).  Here the same
topology is *derived as data*: a flat list of layer records that the HIP engine
walks, plus the table of state_dict names/shapes so that a
real GLIGEN checkpoint loads by name.

Nothing here touches a device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple


@dataclass(frozen=True)
class UNetConfig:
    """Hyper-parameters; defaults are GLIGEN/configs/coco2014.yaml:9-30."""
    image_size: int = 64
    in_channels: int = 4
    model_channels: int = 320
    out_channels: int = 4
    num_res_blocks: int = 2
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    context_dim: int = 768
    pos_in_dim: int = 768        # PositionNet in_dim  (text_grounding_net.py:7)
    pos_out_dim: int = 768       # PositionNet out_dim
    fourier_freqs: int = 8
    max_objs: int = 30           # interface.py:158,425
    # not a hyper-parameter of the network: the packed-weight LAYOUT (gl_unet_config.split_weights).  True stores every matrix as
    # [Whi | Wlo] halves, which the engine's strict mode (option 50: split-fp16 operands, within north_star's tolerance of the fp32
    # reference) needs for its third pass; the default mode of such an engine reads the Whi halves.  Twice the weight bytes.
    split_weights: bool = False

    @property
    def time_embed_dim(self) -> int:
        return self.model_channels * 4

    @property
    def position_dim(self) -> int:
        return self.fourier_freqs * 2 * 4

    @staticmethod
    def from_dict(params: dict) -> "UNetConfig":
        """Accepts the ``config['model']['params']`` dict of a GLIGEN checkpoint."""
        gt = (params.get("grounding_tokenizer") or {}).get("params", {})
        # variants that share parameter names / shapes with this UNet but compute something else must not load silently
        # (load_state_dict(strict=False), interface.py:91, would accept them): openaimodel.py:246-262, attention.py:362-384
        if params.get("fuser_type", "gatedSA") != "gatedSA":
            raise NotImplementedError(f"fuser_type={params.get('fuser_type')!r}: only 'gatedSA' is on the layout-to-image path")
        if int(params.get("transformer_depth", 1)) != 1:
            raise NotImplementedError("transformer_depth != 1 is not supported")
        if params.get("inpaint_mode", False):
            raise NotImplementedError("inpaint_mode checkpoints (9-channel first conv) are not on the layout-to-image path")
        if params.get("grounding_downsampler") is not None:
            raise NotImplementedError("grounding_downsampler is not on the text_layout path")
        if not params.get("use_spatial_transformer", True):
            raise NotImplementedError("use_spatial_transformer=False is not supported")
        return UNetConfig(
            image_size=int(params.get("image_size", 64)),
            in_channels=int(params.get("in_channels", 4)),
            model_channels=int(params.get("model_channels", 320)),
            out_channels=int(params.get("out_channels", 4)),
            num_res_blocks=int(params.get("num_res_blocks", 2)),
            attention_resolutions=tuple(params.get("attention_resolutions", (4, 2, 1))),
            channel_mult=tuple(params.get("channel_mult", (1, 2, 4, 4))),
            num_heads=int(params.get("num_heads", 8)),
            context_dim=int(params.get("context_dim", 768)),
            pos_in_dim=int(gt.get("in_dim", 768)),
            pos_out_dim=int(gt.get("out_dim", 768)),
            fourier_freqs=int(gt.get("fourier_freqs", 8)),
        )


# A small config used by tests (CPU goldens and the GPU whole-model parity test).
# Channels are multiples of 64 (engine GEMM K-tile) and of 32 (GroupNorm32).
TINY = UNetConfig(image_size=16, model_channels=64, num_heads=4)


@dataclass(frozen=True)
class Layer:
    kind: str            # conv_in | res | st | down | up
    prefix: str          # state_dict prefix of the module
    cin: int
    cout: int
    d_head: int = 0


@dataclass
class Block:
    """One TimestepEmbedSequential (openaimodel.py:40-54)."""
    layers: List[Layer] = field(default_factory=list)
    skip_in: int = 0     # channels popped from the skip stack and concatenated (output blocks)


@dataclass
class Plan:
    cfg: UNetConfig
    input_blocks: List[Block]
    middle: Block
    output_blocks: List[Block]
    out_channels_last: int

    def all_layers(self):
        for b in self.input_blocks:
            yield from b.layers
        yield from self.middle.layers
        for b in self.output_blocks:
            yield from b.layers

    def res_layers(self) -> List[Layer]:
        return [l for l in self.all_layers() if l.kind == "res"]

    def st_layers(self) -> List[Layer]:
        return [l for l in self.all_layers() if l.kind == "st"]


def build_plan(cfg: UNetConfig) -> Plan:
    mc, heads = cfg.model_channels, cfg.num_heads
    inputs: List[Block] = [Block([Layer("conv_in", "input_blocks.0.0", cfg.in_channels, mc)])]
    chans = [mc]
    ch, ds = mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            idx = len(inputs)
            layers = [Layer("res", f"input_blocks.{idx}.0", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                layers.append(Layer("st", f"input_blocks.{idx}.1", ch, ch, ch // heads))
            inputs.append(Block(layers))
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            idx = len(inputs)
            inputs.append(Block([Layer("down", f"input_blocks.{idx}.0.op", ch, ch)]))
            chans.append(ch)
            ds *= 2
    middle = Block([
        Layer("res", "middle_block.0", ch, ch),
        Layer("st", "middle_block.1", ch, ch, ch // heads),
        Layer("res", "middle_block.2", ch, ch),
    ])
    outputs: List[Block] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            idx = len(outputs)
            layers = [Layer("res", f"output_blocks.{idx}.0", ch + ich, mc * mult)]
            ch = mc * mult
            if ds in cfg.attention_resolutions:
                layers.append(Layer("st", f"output_blocks.{idx}.1", ch, ch, ch // heads))
            if level and i == cfg.num_res_blocks:
                layers.append(Layer("up", f"output_blocks.{idx}.{len(layers)}.conv", ch, ch))
                ds //= 2
            outputs.append(Block(layers, skip_in=ich))
    return Plan(cfg, inputs, middle, outputs, ch)


def _j(p: str, n: str) -> str:
    return n if p == "" else f"{p}.{n}"


def attn_params(p: str, C: int, kv_dim: int) -> Dict[str, Tuple[int, ...]]:
    return {
        _j(p, "to_q.weight"): (C, C),
        _j(p, "to_k.weight"): (C, kv_dim),
        _j(p, "to_v.weight"): (C, kv_dim),
        _j(p, "to_out.0.weight"): (C, C),
        _j(p, "to_out.0.bias"): (C,),
    }


def ff_params(p: str, C: int) -> Dict[str, Tuple[int, ...]]:
    return {
        _j(p, "net.0.proj.weight"): (8 * C, C),
        _j(p, "net.0.proj.bias"): (8 * C,),
        _j(p, "net.2.weight"): (C, 4 * C),
        _j(p, "net.2.bias"): (C,),
    }


def norm_params(p: str, C: int) -> Dict[str, Tuple[int, ...]]:
    return {_j(p, "weight"): (C,), _j(p, "bias"): (C,)}


def conv_params(p: str, cin: int, cout: int, k: int = 3) -> Dict[str, Tuple[int, ...]]:
    return {_j(p, "weight"): (cout, cin, k, k), _j(p, "bias"): (cout,)}


def res_params(p: str, cin: int, cout: int, te: int) -> Dict[str, Tuple[int, ...]]:
    """ResBlock (openaimodel.py:155-194)."""
    out: Dict[str, Tuple[int, ...]] = {}
    out.update(norm_params(_j(p, "in_layers.0"), cin))
    out.update(conv_params(_j(p, "in_layers.2"), cin, cout))
    out[_j(p, "emb_layers.1.weight")] = (cout, te)
    out[_j(p, "emb_layers.1.bias")] = (cout,)
    out.update(norm_params(_j(p, "out_layers.0"), cout))
    out.update(conv_params(_j(p, "out_layers.3"), cout, cout))
    if cin != cout:
        out.update(conv_params(_j(p, "skip_connection"), cin, cout, 1))
    return out


def fuser_params(p: str, C: int, ctx: int) -> Dict[str, Tuple[int, ...]]:
    """GatedSelfAttentionDense (attention.py:206-224)."""
    out: Dict[str, Tuple[int, ...]] = {_j(p, "alpha_attn"): (), _j(p, "alpha_dense"): ()}
    out[_j(p, "linear.weight")] = (C, ctx)
    out[_j(p, "linear.bias")] = (C,)
    out.update(attn_params(_j(p, "attn"), C, C))
    out.update(ff_params(_j(p, "ff"), C))
    for n in ("norm1", "norm2"):
        out.update(norm_params(_j(p, n), C))
    return out


def rela_params(p: str, C: int, ctx: int) -> Dict[str, Tuple[int, ...]]:
    """RelationCrossAttention (attention.py:284-303)."""
    out: Dict[str, Tuple[int, ...]] = {_j(p, "alpha_attn"): (), _j(p, "alpha_dense"): ()}
    out.update(attn_params(_j(p, "attn"), C, ctx))
    out.update(ff_params(_j(p, "ff"), C))
    for n in ("norm1", "norm2", "norm3"):
        out.update(norm_params(_j(p, n), C))
    return out


def block_params(t: str, C: int, ctx: int) -> Dict[str, Tuple[int, ...]]:
    """BasicTransformerBlock (attention.py:362-384), in module registration order."""
    out: Dict[str, Tuple[int, ...]] = {}
    out.update(attn_params(_j(t, "attn1"), C, C))
    out.update(ff_params(_j(t, "ff"), C))
    out.update(attn_params(_j(t, "attn2"), C, ctx))
    for n in ("norm1", "norm2", "norm3"):
        out.update(norm_params(_j(t, n), C))
    out.update(fuser_params(_j(t, "fuser"), C, ctx))
    out.update(rela_params(_j(t, "rela_fuse"), C, ctx))
    return out


def st_params(p: str, C: int, ctx: int) -> Dict[str, Tuple[int, ...]]:
    """SpatialTransformer (attention.py:405-434)."""
    out: Dict[str, Tuple[int, ...]] = {}
    out.update(norm_params(_j(p, "norm"), C))
    out.update(conv_params(_j(p, "proj_in"), C, C, 1))
    out.update(block_params(_j(p, "transformer_blocks.0"), C, ctx))
    out.update(conv_params(_j(p, "proj_out"), C, C, 1))
    return out


def param_shapes(cfg: UNetConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict name -> shape for the whole UNet, in module registration order."""
    plan = build_plan(cfg)
    te, ctx = cfg.time_embed_dim, cfg.context_dim
    out: Dict[str, Tuple[int, ...]] = {}
    out["time_embed.0.weight"] = (te, cfg.model_channels)
    out["time_embed.0.bias"] = (te,)
    out["time_embed.2.weight"] = (te, te)
    out["time_embed.2.bias"] = (te,)
    for l in plan.all_layers():
        if l.kind in ("conv_in", "down", "up"):
            out.update(conv_params(l.prefix, l.cin, l.cout))
        elif l.kind == "res":
            out.update(res_params(l.prefix, l.cin, l.cout, te))
        elif l.kind == "st":
            out.update(st_params(l.prefix, l.cin, ctx))
        else:
            raise ValueError(l.kind)
    out.update(norm_params("out.0", plan.out_channels_last))
    out.update(conv_params("out.2", plan.out_channels_last, cfg.out_channels))
    out["position_net.null_positive_feature"] = (cfg.pos_in_dim,)
    out["position_net.null_position_feature"] = (cfg.position_dim,)
    out["position_net.linears.0.weight"] = (512, cfg.pos_in_dim + cfg.position_dim)
    out["position_net.linears.0.bias"] = (512,)
    out["position_net.linears.2.weight"] = (512, 512)
    out["position_net.linears.2.bias"] = (512,)
    out["position_net.linears.4.weight"] = (cfg.pos_out_dim, 512)
    out["position_net.linears.4.bias"] = (cfg.pos_out_dim,)
    return out


def count_params(cfg: UNetConfig) -> int:
    n = 0
    for shp in param_shapes(cfg).values():
        k = 1
        for s in shp:
            k *= s
        n += k
    return n


# --------------------------------------------------------------------------- VAE decoder
@dataclass(frozen=True)
class VAEConfig:
    """AutoencoderKL decoder hyper-parameters; defaults are GLIGEN/configs/coco2014.yaml:33-53."""
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    out_ch: int = 3
    embed_dim: int = 4
    scale_factor: float = 0.18215


VAE_TINY = VAEConfig(ch=64, ch_mult=(1, 2), num_res_blocks=1)


def vae_decoder_param_shapes(cfg: VAEConfig) -> Dict[str, Tuple[int, ...]]:
    """state_dict name -> shape of everything AutoencoderKL.decode touches (autoencoder.py:40-44,
    model.py:462-568), in the reference's naming (post_quant_conv.*, decoder.*)."""
    out: Dict[str, Tuple[int, ...]] = {}
    out.update(conv_params("post_quant_conv", cfg.embed_dim, cfg.z_channels, 1))
    nres = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[nres - 1]
    out.update(conv_params("decoder.conv_in", cfg.z_channels, block_in))

    def resnet(p, cin, cout):
        out.update(norm_params(p + ".norm1", cin))
        out.update(conv_params(p + ".conv1", cin, cout))
        out.update(norm_params(p + ".norm2", cout))
        out.update(conv_params(p + ".conv2", cout, cout))
        if cin != cout:
            out.update(conv_params(p + ".nin_shortcut", cin, cout, 1))
    resnet("decoder.mid.block_1", block_in, block_in)
    out.update(norm_params("decoder.mid.attn_1.norm", block_in))
    for n in ("q", "k", "v", "proj_out"):
        out.update(conv_params(f"decoder.mid.attn_1.{n}", block_in, block_in, 1))
    resnet("decoder.mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for i in range(cfg.num_res_blocks + 1):
            resnet(f"decoder.up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            out.update(conv_params(f"decoder.up.{lvl}.upsample.conv", block_in, block_in))
    out.update(norm_params("decoder.norm_out", block_in))
    out.update(conv_params("decoder.conv_out", block_in, cfg.out_ch))
    return out
