"""`train_models_step` of the reference (TransformerTTS/model/traning_steps.py:16-95) over the MI355X trainers: one update of the
autoregressive TTS model, then one ascent step of every mutual-information head on the embeddings that update produced.

Same arguments and the same entries in the returned dict.  `tts_model` is an
autoregressive_training.AutoregressiveTransformerTrainer, `mine_model` a list of mine.MINE / mine.CLUB, `mi_holder` the
reference's dict (use_mine, mine_sep_call, mi_loss, exp_terms, weight_factor, smoothing_factor).
"""
import torch


def train_models_step(tts_mel, tts_phonemes, tts_stop, tts_speaker,
                      mine_mel, mine_phonemes, mine_stop, mine_speaker,
                      tts_model, mine_model, mi_holder,
                      train_text_encoder=True, train_style_encoder=True, train_decoder=True, use_style_loss=False,
                      global_frames=None):
    """global_frames (not in the reference, which is single-GPU): see AutoregressiveTransformerTrainer.train_step."""
    model_out = tts_model.train_step(tts_mel, tts_phonemes, tts_stop, tts_speaker, train_text_encoder=train_text_encoder,
                                     train_style_encoder=train_style_encoder, train_decoder=train_decoder,
                                     use_style_loss=use_style_loss, global_frames=global_frames)
    tts_loss = model_out['tts_loss']
    # the mutual-information term enters the total as a number: mi_holder['mi_loss'] was computed by the heads in the previous
    # step, outside this step's tape, so it moves the reported loss and no gradient (traning_steps.py:52-55)
    mi_prev = torch.as_tensor(mi_holder.get('mi_loss', 0.0), dtype=torch.float32, device=tts_loss.device)
    total_loss = tts_loss + float(mi_holder.get('weight_factor', 0.0)) * torch.clamp(mi_prev, min=0.0)

    if mi_holder.get('use_mine'):
        if mi_holder.get('mine_sep_call'):
            _, mine_processed_mel, _, _ = tts_model.input_reshape(torch.as_tensor(mine_mel), torch.as_tensor(mine_stop))
            mine_gst_out, mine_text_enc_out = tts_model.encode(mine_phonemes, mine_processed_mel, mine_speaker)
        else:
            mine_gst_out, mine_text_enc_out, mine_speaker = model_out['gst_output'], model_out['text_enc_output'], tts_speaker
        mi_loss_gather, exp_terms = {}, None
        for m in mine_model:
            mi_loss, exp_terms = m.call(mine_text_enc_out, mine_gst_out, mine_speaker, mi_holder, train=True)
            mi_loss_gather[m.name + ':' + m.pair_type] = mi_loss
        model_out.update({'mi_loss': dict(mi_loss_gather)})
        mi_holder.update({'mi_loss': sum(mi_loss_gather.values()), 'exp_terms': exp_terms})

    model_out.update({'tts_loss': tts_loss, 'loss': total_loss})
    return model_out
