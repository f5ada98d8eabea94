"""``python main.py generate_embeddings ...`` with the reference's flag set (cli.py:54-175)."""
import click


@click.group()
def main():
    pass


@main.command("generate_embeddings")
@click.option("--encoder", default="vit_h", help="Select the encoder to use")
@click.option("--checkpoint", default="vit_h.pth", help="Select the file to use as checkpoint")
@click.option("--use_sam_checkpoint", is_flag=True, help="Select if the checkpoint is a SAM checkpoint")
@click.option("--compile", is_flag=True, help="Accepted for compatibility (kernels are precompiled HIP)")
@click.option("--directory", default="data/raw/train2017", help="Directory of the images")
@click.option("--batch_size", default=1, help="Batch size")
@click.option("--num_workers", default=0, help="Accepted for compatibility")
@click.option("--outfolder", default="data/processed/embeddings", help="Folder to save the embeddings")
@click.option("--device", default="cuda", help="Device to use for the model")
@click.option("--last_block_dir", default=None, help="Folder to save last transformer block")
@click.option("--custom_preprocess", is_flag=True, help="Whether to use custom resize and normalize")
@click.option("--huggingface", is_flag=True, help="Whether to use huggingface models")
@click.option("--model_name", default="facebook/vit-mae-base", help="Local HF model directory (Only for huggingface models)")
@click.option("--image_resolution", default=480, help="Image resolution for ViT (Only for huggingface models)")
@click.option("--mean_std", default="default", help="Mean and std for normalization (default or standard)")
def generate_embeddings(encoder, checkpoint, use_sam_checkpoint, compile, directory, batch_size, num_workers, outfolder, device,
                        last_block_dir, custom_preprocess, huggingface, model_name, image_resolution, mean_std):
    if huggingface:
        from label_anything.preprocess import preprocess_images_to_embeddings_huggingface
        n = preprocess_images_to_embeddings_huggingface(
            model_name=model_name, directory=directory, batch_size=batch_size, num_workers=num_workers, outfolder=outfolder,
            device=device, compile=compile, image_resolution=image_resolution, custom_preprocess=custom_preprocess,
            mean_std=mean_std)
    else:
        from label_anything.preprocess import preprocess_images_to_embeddings
        n = preprocess_images_to_embeddings(
            encoder_name=encoder, checkpoint=checkpoint, use_sam_checkpoint=use_sam_checkpoint, directory=directory,
            batch_size=batch_size, num_workers=num_workers, outfolder=outfolder, last_block_dir=last_block_dir, device=device,
            compile=compile, custom_preprocess=custom_preprocess)
    click.echo(f"wrote {n} embeddings to {outfolder}")


if __name__ == "__main__":
    main()
